"""K4 SantaLucia scan of a 5 Mb genome (configs[4]): ms, windows/s, written TB/s"""
import sys
import torch
sys.path.insert(0, '.')
from poly_amd import bench_extra
dev = torch.device('cuda:0')
for _ in range(3):
    r = bench_extra.tm_scan(dev)
    print(f"{r['ms']:.4f} ms  {r['windows_per_s']:.3e} windows/s  {r['algorithmic_GBs']:.0f} GB/s", flush=True)
