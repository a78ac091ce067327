"""NeedlemanWunsch throughput, device-resident: n pairs of LA x LB (per-pair B)."""
import sys
import torch
sys.path.insert(0, '.')
from poly_amd import _lib, align, alphabet, matrix, mash
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
LA = LB = int(sys.argv[2]) if len(sys.argv) > 2 else 150
a = alphabet.NewAlphabet(list("-ACGT"))
sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
A = torch.empty(n * LA, dtype=torch.uint8, device=dev)
mash.synth_dna_dev(1, A)
B = A.clone().view(n, LB)
gen = torch.Generator(device=dev); gen.manual_seed(3)
hit = torch.rand(B.shape, device=dev, generator=gen) < 0.05
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
B[hit] = lut[torch.randint(0, 4, (int(hit.sum()),), device=dev, generator=gen)]
B = B.reshape(-1).contiguous()
offA = torch.arange(0, (n + 1) * LA, LA, dtype=torch.int64, device=dev)
offB = torch.arange(0, (n + 1) * LB, LB, dtype=torch.int64, device=dev)
L = _lib.lib()
wb = int(L.polyhip_nw_workspace_bytes(n, LA, LB))
work = torch.empty(wb, dtype=torch.uint8, device=dev)
stride = LA + LB
score = torch.zeros(n, dtype=torch.int64, device=dev)
err, ln = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2))
alnA = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
alnB = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
def f():
    _lib.check(L.polyhip_nw_align_batch_dev(sc.handle(), A.data_ptr(), offA.data_ptr(), n, LA, B.data_ptr(), offB.data_ptr(), LB,
               score.data_ptr(), err.data_ptr(), alnA.data_ptr(), alnB.data_ptr(), ln.data_ptr(), stride, work.data_ptr(), wb, None))
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); f(); f(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 2
print(f"NW: {ms:.2f} ms per {n} pairs of {LA}x{LB} -> {n*LA*LB/ms*1e3:.3e} cell updates/s (workspace {wb/2**30:.2f} GiB), mean score {float(score.double().mean()):.1f}")
