"""K1 A/B on the GPU box: tile pass (POLYHIP_K1_SLABS=0) vs slab pass, same inputs, outputs compared word for word.
usage: python scripts/quick_k1_ab.py [n_reads]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, '.')
from poly_amd import mash
dev = torch.device('cuda:0')
n, L, s = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000, 10_000, 1000
seqs = torch.empty(n * L, dtype=torch.uint8, device=dev)
mash.synth_dna_dev(0xC2, seqs)
offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)


def run(k, slabs, reps=5):
    os.environ["POLYHIP_K1_SLABS"] = "1" if slabs else "0"
    out = torch.zeros((n, s), dtype=torch.int32, device=dev)
    for _ in range(2):
        mash.sketch_batch_dev(seqs, offs, k, s, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        mash.sketch_batch_dev(seqs, offs, k, s, out)
    e1.record()
    torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / reps


for k in (21, 17, 31):
    a, ta = run(k, False)
    b, tb = run(k, True)
    same = bool(torch.equal(a, b))
    km = n * (L - k)
    print(f"k={k}: tiles {ta:.3f} ms ({km / ta * 1e3:.3e} k-mers/s)  slabs {tb:.3f} ms ({km / tb * 1e3:.3e} k-mers/s)  "
          f"speedup {ta / tb:.3f}  identical {same}", flush=True)
    assert same
# ragged / misaligned / short reads: every byte alignment, lengths around the slab and window edges
rng = np.random.default_rng(5)
lens = np.concatenate([rng.integers(0, 3000, 3000), np.arange(990, 1300), np.arange(20, 60), rng.integers(3000, 40000, 300)])
rng.shuffle(lens)
o = np.zeros(len(lens) + 1, np.int64)
o[1:] = np.cumsum(lens)
buf = torch.empty(int(o[-1]) + 64, dtype=torch.uint8, device=dev)
mash.synth_dna_dev(0xAB, buf)
offs2 = torch.from_numpy(o).to(dev)
for k in (21, 17, 31):
    for ss in (1000, 64, 2, 4000):
        res = []
        for slabs in (False, True):
            os.environ["POLYHIP_K1_SLABS"] = "1" if slabs else "0"
            out = torch.full((len(lens), ss), 7, dtype=torch.int32, device=dev)
            mash.sketch_batch_dev(buf, offs2, k, ss, out)
            torch.cuda.synchronize()
            res.append(out)
        assert torch.equal(res[0], res[1]), (k, ss)
print("ragged batches identical")
