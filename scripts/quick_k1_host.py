"""PCIe-inclusive rate of the host-pointer flavour polyhip_mash_sketch_batch (what cgo calls):
pageable numpy buffers in, sketches back in pageable memory."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from poly_amd import mash
n, L, k, s = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000, 10_000, 21, 1000
rng = np.random.default_rng(1)
seqs = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * L, dtype=np.uint8)]
offs = np.arange(0, (n + 1) * L, L, dtype=np.uint64)
out = np.zeros((n, s), dtype=np.uint32)
mash.sketch_batch_packed(seqs[:100 * L], offs[:101], k, s, out[:100])  # warm up (context, module load)
best = 1e9
for _ in range(3):
    t = time.perf_counter()
    mash.sketch_batch_packed(seqs, offs, k, s, out)
    best = min(best, time.perf_counter() - t)
kmers = n * (L - k)
print(f"K1 host flavour: {best*1e3:.1f} ms per {n} reads ({n*L/1e9:.1f} GB in, {n*s*4/1e9:.1f} GB out) -> "
      f"{kmers/best:.3e} k-mers/s, {(n*L + n*s*4)/best/1e9:.1f} GB/s over PCIe")
print("checksum", int(out.sum(dtype=np.uint64)) & 0xFFFFFFFF)
