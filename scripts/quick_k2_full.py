"""config 3 whole matrix on one GPU (100k x 100k, one index): ms of one polyhip_mash_shared_counts_dev call and of the join
alone over an index that is already there (POLYHIP_LIB selects the build)"""
import os, sys
import torch
sys.path.insert(0, '.')
from poly_amd import bench_extra, mash
dev = torch.device('cuda:0')
s = 1000
sk = bench_extra.family_sketches(dev, 1000, 100, 10_000, 21, s, 0xC3)
N = sk.shape[0]
counts = torch.empty((N, N), dtype=torch.int16, device=dev)
work = torch.empty(mash.shared_counts_workspace_bytes(N, s, N, s), dtype=torch.uint8, device=dev)
def t(fn, reps=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms_o = t(lambda: mash.shared_counts_dev(sk, sk, counts, work))
mash.index_build_dev(sk, work)
ms_j = t(lambda: mash.shared_counts_reuse_dev(sk, sk, counts, work))
print(f"{os.environ.get('POLYHIP_LIB', 'default'):32s} full one-shot {ms_o:.3f} ms  join {ms_j:.3f} ms  diag ok {bool((counts.diagonal() == s).all())} nonzero {int((counts != 0).sum())}")
