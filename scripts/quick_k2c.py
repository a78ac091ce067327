import os, sys
import torch
sys.path.insert(0, '.')
from poly_amd import bench_extra, mash
dev = torch.device('cuda:0')
s = 1000
sk = bench_extra.family_sketches(dev, 1000, 100, 10_000, 21, s, 0xC3)
N = sk.shape[0]
nrows = N // 8
X = sk[:nrows]
counts = torch.full((nrows, N), -1, dtype=torch.int16, device=dev)
work = torch.empty(mash.shared_counts_workspace_bytes(nrows, s, N, s), dtype=torch.uint8, device=dev)
mash.index_build_dev(sk, work)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for abl, what in ((0, "full"), (1, "no flush"), (2, "no LDS atomics"), (3, "no flush, no atomics"), (4, "no bucket walk"), (5, "no walk, no flush (row prep + zero only)")):
    os.environ["POLYHIP_K2_ABL"] = str(abl)
    print(f"abl {abl} {what:45s} {t(lambda: mash.shared_counts_reuse_dev(X, sk, counts, work)):.3f} ms", flush=True)
