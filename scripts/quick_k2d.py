"""index build + join timing of one library build (POLYHIP_LIB selects it), config 3 row block; prints a checksum"""
import os, sys
import torch
sys.path.insert(0, '.')
from poly_amd import bench_extra, mash
dev = torch.device('cuda:0')
s = 1000
sk = bench_extra.family_sketches(dev, 1000, 100, 10_000, 21, s, 0xC3)
N = sk.shape[0]
nrows = N // 8
X = sk[:nrows]
counts = torch.full((nrows, N), -1, dtype=torch.int16, device=dev)
work = torch.empty(mash.shared_counts_workspace_bytes(nrows, s, N, s), dtype=torch.uint8, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms_i = t(lambda: mash.index_build_dev(sk, work))
ms_j = t(lambda: mash.shared_counts_reuse_dev(X, sk, counts, work))
ms_o = t(lambda: mash.shared_counts_dev(X, sk, counts, work))
print(f"{os.environ.get('POLYHIP_LIB', 'default'):32s} index {ms_i:.3f} ms  join {ms_j:.3f} ms  one-shot {ms_o:.3f} ms  checksum {int(counts.to(torch.int64).sum())} {int((counts != 0).sum())}")
