"""K1 rate for several k (compile-time instantiations 17 / 21 / 31 vs the runtime-k kernel)."""
import sys, torch
sys.path.insert(0, '.')
from poly_amd import mash
dev = torch.device('cuda:0')
n, L, s = 100_000, 10_000, 1000
seqs = torch.empty(n * L, dtype=torch.uint8, device=dev); mash.synth_dna_dev(0xC2, seqs)
offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
out = torch.zeros((n, s), dtype=torch.int32, device=dev)
for k in (12, 16, 17, 20, 21, 24, 25, 27, 31, 32, 51):
    mash.sketch_batch_dev(seqs, offs, k, s, out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): mash.sketch_batch_dev(seqs, offs, k, s, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"k={k:3d}: {ms:7.3f} ms per {n} reads -> {n*(L-k)/ms*1e3:.3e} k-mers/s")
