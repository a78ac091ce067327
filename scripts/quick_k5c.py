"""K5: where the wave-per-sequence kernel hands over to the workgroup kernel (POLYHIP_K5_WAVE_MAX), 0.5 GB of random DNA"""
import os
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
for L in (3000, 4000, 6000, 8000):
    n = tot // L
    offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    rot = torch.zeros(n, dtype=torch.int64, device=dev)
    res = []
    for wm in ("8192", "2048"):
        os.environ["POLYHIP_K5_WAVE_MAX"] = wm
        res.append(_time(lambda: seqhash.least_rotation_batch_dev(rnd, offs, L, rot, out), 10))
    print(f"L={L}: wave kernel {res[0]:.3f} ms, workgroup kernel {res[1]:.3f} ms", flush=True)
