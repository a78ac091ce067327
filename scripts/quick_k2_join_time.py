"""K2 join alone at config 3 (12,500 x 100,000 against a prebuilt index), for A/B of library variants (POLYHIP_LIB=...)"""
import sys
import torch
sys.path.insert(0, '.')
from poly_amd import bench_extra, mash
from poly_amd.bench_extra import _time
dev = torch.device('cuda:0')
s = 1000
sk = bench_extra.family_sketches(dev, 1000, 100, 10_000, 21, s, 0xC3)
N = sk.shape[0]
nrows = N // 8
X = sk[:nrows]
counts = torch.full((nrows, N), -1, dtype=torch.int16, device=dev)
work = torch.empty(mash.shared_counts_workspace_bytes(nrows, s, N, s), dtype=torch.uint8, device=dev)
mash.index_build_dev(sk, work)
ms = _time(lambda: mash.shared_counts_reuse_dev(X, sk, counts, work), 20)
print(f"join {ms:.3f} ms  nonzero {int((counts != 0).sum())}  diag {bool((counts[:, :nrows].diagonal() == s).all())}")
