import sys, torch
sys.path.insert(0, '.')
from poly_amd import bench_extra
dev = torch.device('cuda:0')
for _ in range(3):
    r = bench_extra.hashing(dev)
    print(f"seqhash {r['ms']:.4f} ms  {r['sequences_per_s']:.3e} seq/s", flush=True)
