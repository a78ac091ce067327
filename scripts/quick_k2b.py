"""K2 on the config-3 set (1000 families x 100 copies, s=1000): dense join vs sparse join, one-shot vs reused index,
1/8 row block and the full matrix; results compared between the join kinds."""
import os, sys
import torch
sys.path.insert(0, '.')
from poly_amd import bench_extra, mash
dev = torch.device('cuda:0')
s = 1000
sk = bench_extra.family_sketches(dev, 1000, 100, 10_000, 21, s, 0xC3)
N = sk.shape[0]


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ref = {}
for kind in ("dense", "sparse"):
    os.environ["POLYHIP_K2_DENSE"] = "1" if kind == "dense" else "0"
    for nrows, tag in ((N // 8, "1/8 block"), (N, "full")):
        if nrows == N and kind == "sparse" and len(sys.argv) > 1:
            continue
        X = sk[:nrows]
        counts = torch.full((nrows, N), -1, dtype=torch.int16, device=dev)
        work = torch.empty(mash.shared_counts_workspace_bytes(nrows, s, N, s), dtype=torch.uint8, device=dev)
        ms = t(lambda: mash.shared_counts_dev(X, sk, counts, work), 3 if nrows == N else 5)
        ms_idx = t(lambda: mash.index_build_dev(sk, work))
        ms_re = t(lambda: mash.shared_counts_reuse_dev(X, sk, counts, work), 3 if nrows == N else 5)
        pairs = nrows * N
        chk = (int(counts.to(torch.int64).sum()), int((counts != 0).sum()))
        print(f"{kind:6s} {tag:9s}: one-shot {ms:.3f} ms ({pairs / ms * 1e3:.3e} pairs/s, {pairs * 2 / ms * 1e3 / 1e9:.0f} GB/s u16)  "
              f"index {ms_idx:.3f} ms  join-only {ms_re:.3f} ms ({pairs * 2 / ms_re * 1e3 / 1e9:.0f} GB/s)  mode {mash.shared_counts_mode(work)[:4]} "
              f"sum/nonzero {chk}", flush=True)
        if tag in ref:
            assert ref[tag] == chk, (ref[tag], chk)
            assert torch.equal(counts, refc[tag]) if tag == "1/8 block" else True
        else:
            ref[tag] = chk
            if tag == "1/8 block":
                refc = {tag: counts.clone()}
        del counts, work
