"""K5 on random DNA: index only vs index + rotated copy, at several sequence lengths (0.5 GB of sequence each)"""
import sys
import torch
sys.path.insert(0, '.')
from poly_amd import mash, seqhash
from poly_amd.bench_extra import _time
dev = torch.device('cuda:0')
tot = 500_000_000
rnd = torch.empty(tot, dtype=torch.uint8, device=dev)
mash.synth_dna_dev(0x5EED, rnd)
out = torch.zeros(tot, dtype=torch.uint8, device=dev)
for L in (200, 1000, 5000, 20000, 100000):
    n = tot // L
    offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    rot = torch.zeros(n, dtype=torch.int64, device=dev)
    a = _time(lambda: seqhash.least_rotation_batch_dev(rnd, offs, L, rot, None), 10)
    b = _time(lambda: seqhash.least_rotation_batch_dev(rnd, offs, L, rot, out), 10)
    print(f"L={L}: index only {a:.3f} ms ({n * L / a / 1e6:.0f} GB/s read), with rotated copy {b:.3f} ms ({2 * n * L / b / 1e6:.0f} GB/s moved)", flush=True)
