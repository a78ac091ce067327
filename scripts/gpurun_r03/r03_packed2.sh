#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in 1 2 3 4 8; do
POLYHIP_SW_HOST_CHUNKS=$c python - <<'PY'
import os, sys, torch, numpy as np
sys.path.insert(0, '.')
from poly_amd import bench_extra, align, alphabet, matrix, workloads, _lib
from poly_amd.bench_extra import _wall
dev = torch.device('cuda:0')
a = alphabet.NewAlphabet(list("-ACGT"))
sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
n, LA, LB = 1_000_000, 150, 5000
B, A = workloads.config4_reads(n, LA, LB, device=dev)
hA, hB = A.reshape(-1).cpu().numpy(), B.cpu().numpy()
del A, B
offA = np.arange(0, (n + 1) * LA, LA, dtype=np.uint64)
L_ = _lib.lib()
o_score = np.zeros(n, np.int64); o_ea, o_eb, o_er = (np.zeros(n, np.uint32) for _ in range(3))
cap = n * 200
p_a, p_b, p_off = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8), np.zeros(n + 1, np.uint64)
def f():
    _lib.check(L_.polyhip_sw_align_batch_packed(sc.handle(), hA.ctypes.data, offA.ctypes.data, n, hB.ctypes.data, None, LB,
               o_score.ctypes.data, o_ea.ctypes.data, o_eb.ctypes.data, o_er.ctypes.data, p_a.ctypes.data, p_b.ctypes.data, p_off.ctypes.data, cap))
print("chunks", os.environ["POLYHIP_SW_HOST_CHUNKS"], "packed ms", round(_wall(f, 3, 1), 2), flush=True)
PY
done
