"""Times K3 score pass + traceback at BASELINE config 4 size on the GPU."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from poly_amd import align, alphabet, matrix, mash
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
LA = int(sys.argv[2]) if len(sys.argv) > 2 else 150
LB = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
a = alphabet.NewAlphabet(list("-ACGT"))
sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
# windows of the reference with 5 % substitutions + 1 % indels (the input bench.py times)
from poly_amd import workloads
B, A = workloads.config4_reads(n, LA, LB, device=dev)
A = A.reshape(-1).contiguous()
offA = torch.arange(0, (n + 1) * LA, LA, dtype=torch.int64, device=dev)
score = torch.zeros(n, dtype=torch.int64, device=dev)
ea, eb, er, ln = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4))
work = torch.empty(align.sw_workspace_bytes(sc, n, LA, LB, True), dtype=torch.uint8, device=dev)
stride = align.sw_traceback_stride(sc, LA, LB)
tbw = torch.empty(align.sw_traceback_workspace_bytes(sc, n, LA, LB), dtype=torch.uint8, device=dev)
alnA = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
alnB = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
def t(f, R=2):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / R
ms1 = t(lambda: align.sw_batch_dev(sc, A, offA, LA, B, None, LB, score, ea, eb, er, work))
ms2 = t(lambda: align.sw_traceback_dev(sc, A, offA, LA, B, None, LB, ea, eb, er, alnA, alnB, ln, tbw, score_t=score))
cells = n * LA * LB
print(f"K3 score: {ms1:.2f} ms ({cells/ms1*1e3:.3e} CUPS); traceback: {ms2:.2f} ms (workspace {tbw.numel()/2**30:.1f} GiB, stride {stride}); "
      f"both: {cells/(ms1+ms2)*1e3:.3e} CUPS; mean score {float(score.double().mean()):.1f} mean aln len {float(ln.double().mean()):.1f}")
