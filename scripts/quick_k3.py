"""Times K3 (SW score pass) at BASELINE config 4 size on the GPU."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from poly_amd import align, alphabet, matrix, mash
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
LA, LB = 150, 5000
a = alphabet.NewAlphabet(list("-ACGT"))
sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
A = torch.empty(n * LA, dtype=torch.uint8, device=dev)
mash.synth_dna_dev(0xC4 + 1, A)
B = torch.empty(LB, dtype=torch.uint8, device=dev)
mash.synth_dna_dev(0xC4, B)
offA = torch.arange(0, (n + 1) * LA, LA, dtype=torch.int64, device=dev)
score = torch.zeros(n, dtype=torch.int64, device=dev)
ea = torch.zeros(n, dtype=torch.int32, device=dev)
eb = torch.zeros(n, dtype=torch.int32, device=dev)
er = torch.zeros(n, dtype=torch.int32, device=dev)
wb = align.sw_workspace_bytes(sc, n, LA, LB, True)
work = torch.empty(wb, dtype=torch.uint8, device=dev)
def step():
    align.sw_batch_dev(sc, A, offA, LA, B, None, LB, score, ea, eb, er, work)
step(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
R = 3
e0.record()
for _ in range(R):
    step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / R
cells = n * LA * LB
print(f"K3: {ms:.3f} ms per {n} pairs -> {cells/ms*1e3:.3e} CUPS path={align.last_path()} maxscore={int(score.max())}")
