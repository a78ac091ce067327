"""host-pointer seqhash / least rotation / SantaLucia batch at several chunk sizes of the two-slot pipeline"""
import os, sys, time
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
o_h, o_e = np.zeros(nq * 72, np.uint8), np.zeros(nq, np.uint32)
o_rot, o_seq = np.zeros(nq, np.uint64), np.zeros(nq * Lq, np.uint8)
def wall(f, reps=5):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]
for mb in (16, 32, 64, 128, 256, 512):
    os.environ['POLYHIP_HOST_CHUNK_MB'] = str(mb)
    a = wall(lambda: _lib.check(L_.polyhip_seqhash_batch(hq.ctypes.data, oq.ctypes.data, nq, 0, 1, 1, o_h.ctypes.data, o_e.ctypes.data)))
    b = wall(lambda: _lib.check(L_.polyhip_least_rotation_batch(hq.ctypes.data, oq.ctypes.data, nq, o_rot.ctypes.data, o_seq.ctypes.data)))
    print(f"chunk {mb:4d} MB: seqhash {a:6.2f} ms  least_rotation(+rotated back) {b:6.2f} ms", flush=True)
