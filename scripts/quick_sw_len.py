"""score pass (and traceback) of n reads of LA bp against one LB bp reference -- profiler driver for the long-read legs:
    python scripts/quick_sw_len.py 400000 250 5000"""
import sys, torch
sys.path.insert(0, '.')
from poly_amd import align, alphabet, matrix, workloads
n, LA, LB = (int(x) for x in sys.argv[1:4])
dev = torch.device('cuda:0')
a = alphabet.NewAlphabet(list("-ACGT"))
sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
B, A = workloads.config4_reads(n, LA, LB, first=0, device=dev)
A = A.reshape(-1).contiguous()
offA = torch.arange(0, (n + 1) * LA, LA, dtype=torch.int64, device=dev)
work = torch.empty(align.sw_workspace_bytes(sc, n, LA, LB, True), dtype=torch.uint8, device=dev)
score = torch.zeros(n, dtype=torch.int64, device=dev)
ea, eb, er = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3))
for _ in range(3):
    align.sw_batch_dev(sc, A, offA, LA, B, None, LB, score, ea, eb, er, work)
torch.cuda.synchronize()
print("path", align.last_path(), "lanes", align.last_packed_lanes())
