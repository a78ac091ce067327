"""Times K5 (least rotation) on a batch of plasmid-scale circular sequences."""
import sys
import torch
sys.path.insert(0, '.')
from poly_amd import seqhash, mash
dev = torch.device('cuda:0')
n, L = 100_000, 5_000
seqs = torch.empty(n * L, dtype=torch.uint8, device=dev)
mash.synth_dna_dev(0x5EED, seqs)
offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
rot = torch.zeros(n, dtype=torch.int64, device=dev)
out = torch.zeros_like(seqs)
seqhash.least_rotation_batch_dev(seqs, offs, L, rot, out); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
R = 5
e0.record()
for _ in range(R):
    seqhash.least_rotation_batch_dev(seqs, offs, L, rot, out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / R
print(f"K5: {ms:.3f} ms per {n} x {L} B -> {n*L/ms*1e3:.3e} bases/s, {(2*n*L+8*n)/ms*1e3/1e9:.1f} GB/s algorithmic")
