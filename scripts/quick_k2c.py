"""K2 at config 3 (12,500 x 100,000 row block) for the profiler: index build once, then the join-only call and the one-shot call"""
import sys
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
for _ in range(5):
    mash.shared_counts_reuse_dev(X, sk, counts, work)
torch.cuda.synchronize()
for _ in range(3):
    mash.shared_counts_dev(X, sk, counts, work)
torch.cuda.synchronize()
print("nonzero", int((counts != 0).sum()))
