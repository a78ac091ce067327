import sys, torch
sys.path.insert(0, '.')
from poly_amd import mash, seqhash
dev = torch.device('cuda:0')
n, L = 100_000, 5000
offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
rot = torch.zeros(n, dtype=torch.int64, device=dev)
out = torch.zeros(n * L, dtype=torch.uint8, device=dev)
rnd = torch.empty(n * L, dtype=torch.uint8, device=dev)
mash.synth_dna_dev(0x5EED, rnd)
for _ in range(6):
    seqhash.least_rotation_batch_dev(rnd, offs, L, rot, out)
torch.cuda.synchronize()
