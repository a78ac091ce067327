import sys, torch
sys.path.insert(0, '.')
from poly_amd import bench_extra, mash
dev = torch.device('cuda:0')
s = 1000
sk = bench_extra.family_sketches(dev, 1000, 100, 10_000, 21, s, 0xC3)
N = sk.shape[0]; nrows = N // 8
counts = torch.empty((nrows, N), dtype=torch.int16, device=dev)
work = torch.empty(mash.shared_counts_workspace_bytes(nrows, s, N, s), dtype=torch.uint8, device=dev)
mash.index_build_dev(sk, work)
print("join only ms", bench_extra._time(lambda: mash.shared_counts_reuse_dev(sk[:nrows], sk, counts, work), 10))
