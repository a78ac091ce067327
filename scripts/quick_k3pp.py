"""SW score pass + traceback with PER-PAIR B (reads vs reads): n pairs of L x L."""
import sys
import torch
sys.path.insert(0, '.')
from poly_amd import align, alphabet, matrix, mash
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
a = alphabet.NewAlphabet(list("-ACGT"))
sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
A = torch.empty(n * L, dtype=torch.uint8, device=dev)
mash.synth_dna_dev(1, A)
B = A.clone().view(n, L)
gen = torch.Generator(device=dev); gen.manual_seed(3)
hit = torch.rand(B.shape, device=dev, generator=gen) < 0.05
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
B[hit] = lut[torch.randint(0, 4, (int(hit.sum()),), device=dev, generator=gen)]
B = B.reshape(-1).contiguous()
off = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
score = torch.zeros(n, dtype=torch.int64, device=dev)
ea, eb, er, ln = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4))
work = torch.empty(align.sw_workspace_bytes(sc, n, L, L, False), dtype=torch.uint8, device=dev)
stride = align.sw_traceback_stride(sc, L, L)
tbw = torch.empty(align.sw_traceback_workspace_bytes(sc, n, L, L), dtype=torch.uint8, device=dev)
alnA = torch.zeros((n, stride), dtype=torch.uint8, device=dev); alnB = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
def t(f, R=2):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / R
ms1 = t(lambda: align.sw_batch_dev(sc, A, off, L, B, off, L, score, ea, eb, er, work))
p = align.last_path()
ms2 = t(lambda: align.sw_traceback_dev(sc, A, off, L, B, off, L, ea, eb, er, alnA, alnB, ln, tbw, score_t=score))
cells = n * L * L
print(f"per-pair B: score {ms1:.2f} ms ({cells/ms1*1e3:.3e} CUPS, path {p}); traceback {ms2:.2f} ms (path {align.sw_traceback_last_path()}); mean score {float(score.double().mean()):.1f}")
