"""polyhip_sw_align_batch (host pointers, aligned strings) on config 4: wall time per POLYHIP_SW_HOST_CHUNKS setting"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from poly_amd import _lib, align, alphabet, matrix, workloads
dev = torch.device('cuda:0')
a = alphabet.NewAlphabet(list("-ACGT"))
sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
n, LA, LB = 1_000_000, 150, 5000
B, A = workloads.config4_reads(n, LA, LB, device=dev)
hA, hB = A.reshape(-1).cpu().numpy(), B.cpu().numpy()
del A, B
offA = np.arange(0, (n + 1) * LA, LA, dtype=np.uint64)
L_ = _lib.lib()
stride = align.sw_traceback_stride(sc, LA, LB)
o_score, o_len = np.zeros(n, np.int64), np.zeros(n, np.uint32)
o_ea, o_eb, o_er = (np.zeros(n, np.uint32) for _ in range(3))
o_alnA, o_alnB = (np.zeros((n, stride), np.uint8) for _ in range(2))
def run():
    _lib.check(L_.polyhip_sw_align_batch(sc.handle(), hA.ctypes.data, offA.ctypes.data, n, hB.ctypes.data, None, LB,
                                         o_score.ctypes.data, o_ea.ctypes.data, o_eb.ctypes.data, o_er.ctypes.data,
                                         o_alnA.ctypes.data, o_alnB.ctypes.data, o_len.ctypes.data, stride))
for setting in (sys.argv[1:] or ["1", "", "2", "4", "8"]):
    if setting:
        os.environ["POLYHIP_SW_HOST_CHUNKS"] = setting
    else:
        os.environ.pop("POLYHIP_SW_HOST_CHUNKS", None)
    run()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); run(); ts.append(time.perf_counter() - t0)
    print(f"chunks={setting or 'default'}: {1e3 * sorted(ts)[len(ts) // 2]:.1f} ms  checksum {int(o_score.sum())} {int(o_len.sum())} stride {stride}")
