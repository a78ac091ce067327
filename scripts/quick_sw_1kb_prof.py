"""profiler driver: bench_extra's 1 kb leg once -- score pass, traceback, and the one-call form (80k x 1000 bp vs 5 kb)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poly_amd import align, alphabet, matrix, workloads

dev = torch.device("cuda:0")
a = alphabet.NewAlphabet(list("-ACGT"))
sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
n, LA, LB = 80_000, 1000, 5000
B, A = workloads.config4_reads(n, LA, LB, first=0, device=dev)
A = A.reshape(-1).contiguous()
offA = torch.arange(0, (n + 1) * LA, LA, dtype=torch.int64, device=dev)
score = torch.zeros(n, dtype=torch.int64, device=dev)
ea, eb, er, ln = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4))
work = torch.empty(align.sw_workspace_bytes(sc, n, LA, LB, True), dtype=torch.uint8, device=dev)
stride = align.sw_traceback_stride(sc, LA, LB)
tbw = torch.empty(align.sw_traceback_workspace_bytes(sc, n, LA, LB), dtype=torch.uint8, device=dev)
alnA = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
alnB = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
for _ in range(2):
    align.sw_batch_dev(sc, A, offA, LA, B, None, LB, score, ea, eb, er, work)
    align.sw_traceback_dev(sc, A, offA, LA, B, None, LB, ea, eb, er, alnA, alnB, ln, tbw, score_t=score)
    align.sw_align_dev(sc, A, offA, LA, B, None, LB, score, ea, eb, er, alnA, alnB, ln, work, tbw)
torch.cuda.synchronize()
print("paths", align.last_path(), align.sw_traceback_last_path(), "mean length", float(ln.double().mean()))
