import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from poly_amd import mash, devices
import oracle as orc
dev = torch.device('cuda:0')
n, L, k, s = 200_000, 10_000, 21, 1000
d = torch.empty(n * L, dtype=torch.uint8, device=dev)
mash.synth_dna_dev(0xC2, d)
host = d.cpu().numpy(); del d
offs = np.arange(0, (n + 1) * L, L, dtype=np.uint64)
sk = np.zeros((n, s), dtype=np.uint32)
def wall(f, reps=5):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]
ms = wall(lambda: mash.sketch_batch_packed(host, offs, k, s, out=sk))
print(f"K1 host call 200k x 10 kb: {ms:.2f} ms = {n*(L-k)/ms*1e3:.3e} k-mers/s, {(n*L+4*n*s)/ms*1e3/1e9:.1f} GB/s over PCIe (up + down)")
want = orc.mash_sketch_batch(host[:50 * L], offs[:51], k, s)
assert (sk[:50] == want).all() and (sk[-1] == orc.mash_sketch_batch(host[-L:], offs[:2], k, s)[0]).all()
print("spot check ok")
