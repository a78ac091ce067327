"""K2 index build at config 3, timed per kernel with HIP events around the whole build (for A/B of library variants:
POLYHIP_LIB=poly_amd/libpolyhip_<tag>.so); prints the build's ms and its info"""
import sys
import torch
sys.path.insert(0, '.')
from poly_amd import bench_extra, mash
from poly_amd.bench_extra import _time
dev = torch.device('cuda:0')
s = 1000
sk = bench_extra.family_sketches(dev, 1000, 100, 10_000, 21, s, 0xC3)
N = sk.shape[0]
work = torch.empty(mash.shared_counts_workspace_bytes(N // 8, s, N, s), dtype=torch.uint8, device=dev)
ms = _time(lambda: mash.index_build_dev(sk, work), 20)
print(f"index {ms:.3f} ms  {mash.index_build_info(work)}")
