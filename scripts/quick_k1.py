import torch, time, sys
sys.path.insert(0, '.')
from poly_amd import mash
dev = torch.device('cuda:0')
n, L, k, s = 100_000, 10_000, 21, 1000
seqs = torch.empty(n*L, dtype=torch.uint8, device=dev)
mash.synth_dna_dev(0xC2, seqs)
offs = torch.arange(0, (n+1)*L, L, dtype=torch.int64, device=dev)
out = torch.zeros((n, s), dtype=torch.int32, device=dev)
for _ in range(2):
    mash.sketch_batch_dev(seqs, offs, k, s, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
R = 5
for _ in range(R):
    mash.sketch_batch_dev(seqs, offs, k, s, out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)/R
kmers = n*(L-k)
print(f"K1: {ms:.3f} ms per {n} reads -> {kmers/ms*1e3:.3e} kmers/s, {kmers/ms*1e3*1.403/1e9:.1f} GB/s algorithmic")
