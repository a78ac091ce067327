"""K2 when every sketch has thousands of relatives (rows overflow the LDS accumulator)."""
import sys, torch
sys.path.insert(0, '.')
from poly_amd import mash, bench_extra
dev = torch.device('cuda:0')
nfam, copies = int(sys.argv[1]), int(sys.argv[2])
s = 1000
sk = bench_extra.family_sketches(dev, nfam, copies, 10_000, 21, s, seed=0xC3)
N = sk.shape[0]; nrows = min(N, int(sys.argv[3]) if len(sys.argv) > 3 else N // 8)
X = sk[:nrows]
counts = torch.empty((nrows, N), dtype=torch.int16, device=dev)
work = torch.empty(mash.shared_counts_workspace_bytes(nrows, s, N, s), dtype=torch.uint8, device=dev)
mash.shared_counts_dev(X, sk, counts, work); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); mash.shared_counts_dev(X, sk, counts, work); e1.record(); torch.cuda.synchronize()
print(f"{nfam} families x {copies}: {nrows} x {N} pairs in {e0.elapsed_time(e1):.2f} ms; mode {mash.shared_counts_mode(work)}; nonzero {int((counts != 0).sum())}")
