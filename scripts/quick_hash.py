import sys, torch
sys.path.insert(0, '.')
from poly_amd import bench_extra
dev = torch.device('cuda:0')
print(bench_extra.hashing(dev))
print(bench_extra.rotation(dev))
