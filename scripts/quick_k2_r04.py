"""K2 at config 3, round 4: the dense join with a row of <= 1024 hashes in registers (default) against the row staged in
LDS (POLYHIP_K2_REGROW=0), compact and 8-byte items; index / join / one-shot / full-matrix times, counts compared."""
import os
import sys
import torch
sys.path.insert(0, '.')
from poly_amd import bench_extra, mash
from poly_amd.bench_extra import _time
dev = torch.device('cuda:0')
s = 1000
sk = bench_extra.family_sketches(dev, 1000, 100, 10_000, 21, s, 0xC3)
N = sk.shape[0]
nrows = N // 8
X = sk[:nrows]
counts = torch.full((nrows, N), -1, dtype=torch.int16, device=dev)
work = torch.empty(mash.shared_counts_workspace_bytes(nrows, s, N, s), dtype=torch.uint8, device=dev)
res = {}
for tag, env in (("reg", {}), ("unsliced", {"POLYHIP_K2_SLICED": "0"}), ("staged", {"POLYHIP_K2_REGROW": "0"}), ("reg-wide", {"POLYHIP_K2_COMPACT": "0"}),
                 ("unsliced-wide", {"POLYHIP_K2_COMPACT": "0", "POLYHIP_K2_SLICED": "0"})):
    for k in ("POLYHIP_K2_COMPACT", "POLYHIP_K2_REGROW", "POLYHIP_K2_SLICED"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ms_index = _time(lambda: mash.index_build_dev(sk, work), 10)
    torch.cuda.synchronize()
    fmt = mash.index_item_bytes(work)
    ms_join = _time(lambda: mash.shared_counts_reuse_dev(X, sk, counts, work), 10)
    ms_one = _time(lambda: mash.shared_counts_dev(X, sk, counts, work), 10)
    torch.cuda.synchronize()
    res[tag] = counts.clone()
    print(f"{tag}: item bytes {fmt}  index {ms_index:.3f} ms  join {ms_join:.3f} ms  one-shot {ms_one:.3f} ms  nonzero {int((counts != 0).sum())}", flush=True)
print("counts equal:", all(bool(torch.equal(res["reg"], r)) for r in res.values()))
for k in ("POLYHIP_K2_COMPACT", "POLYHIP_K2_REGROW", "POLYHIP_K2_SLICED"):
    os.environ.pop(k, None)
if len(sys.argv) > 1 and sys.argv[1] == "full":
    del res
    cf = torch.empty((N, N), dtype=torch.int16, device=dev)
    wf = torch.empty(mash.shared_counts_workspace_bytes(N, s, N, s), dtype=torch.uint8, device=dev)
    for tag, env in (("reg", None), ("staged", "0")):
        if env is None:
            os.environ.pop("POLYHIP_K2_REGROW", None)
        else:
            os.environ["POLYHIP_K2_REGROW"] = env
        ms_full = _time(lambda: mash.shared_counts_dev(sk, sk, cf, wf), 5)
        print(f"{tag}: full {N} x {N}: {ms_full:.2f} ms  diag ok {bool((cf.diagonal() == s).all())}", flush=True)
