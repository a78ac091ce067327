import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from poly_amd import _lib, mash
L_ = _lib.lib()
dev = torch.device('cuda:0')
nq, Lq = 100_000, 5000
dq = torch.empty(nq * Lq, dtype=torch.uint8, device=dev)
mash.synth_dna_dev(0x5EED, dq)
hq = dq.cpu().numpy(); del dq
oq = np.arange(0, (nq + 1) * Lq, Lq, dtype=np.uint64)
o_rot, o_seq = np.zeros(nq, np.uint64), np.ones(nq * Lq, np.uint8)
def wall(f, reps=7):
    f(); f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2], min(ts)
print("index only      ", wall(lambda: _lib.check(L_.polyhip_least_rotation_batch(hq.ctypes.data, oq.ctypes.data, nq, o_rot.ctypes.data, None))))
print("index + rotated ", wall(lambda: _lib.check(L_.polyhip_least_rotation_batch(hq.ctypes.data, oq.ctypes.data, nq, o_rot.ctypes.data, o_seq.ctypes.data))))
