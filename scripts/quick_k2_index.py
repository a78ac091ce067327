"""K2 index build alone at config 3's size (100,000 sketches of 1000), for the profiler"""
import sys
import torch
sys.path.insert(0, '.')
from poly_amd import bench_extra, mash
dev = torch.device('cuda:0')
s = 1000
sk = bench_extra.family_sketches(dev, 1000, 100, 10_000, 21, s, 0xC3)
N = sk.shape[0]
work = torch.empty(mash.shared_counts_workspace_bytes(N // 8, s, N, s), dtype=torch.uint8, device=dev)
for _ in range(6):
    mash.index_build_dev(sk, work)
torch.cuda.synchronize()
