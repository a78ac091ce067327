"""K4 scan at genome lengths that make the plane stride odd / even (store alignment), and with fewer planes"""
import sys
import torch
sys.path.insert(0, '.')
from poly_amd import bench_extra
dev = torch.device('cuda:0')
for n, Lmin, Lmax in ((5_000_000, 18, 30), (5_000_001, 18, 30), (5_000_001, 18, 18), (5_000_001, 24, 30)):
    r = bench_extra.tm_scan(dev, n, Lmin, Lmax)
    print(f"n={n} L={Lmin}..{Lmax}: {r['ms']:.4f} ms  {r['windows_per_s']:.3e} windows/s  {r['algorithmic_GBs']:.0f} GB/s", flush=True)
