"""reads of 250 / 500 / 1000 bp against one 5 kb reference: the packed multi-lane score pass on 64 rows per lane (four waves per
SIMD, POLYHIP_SW_TILE64=1) against 128 / 152 rows per lane (two waves, =0) -- every pair's four outputs equal, times side by side"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poly_amd import align, alphabet, matrix, workloads

dev = torch.device("cuda:0")
a = alphabet.NewAlphabet(list("-ACGT"))
sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
for n, LA, LB in ((400_000, 250, 5000), (160_000, 500, 5000), (80_000, 1000, 5000), (100_000, 700, 3000)):
    B, A = workloads.config4_reads(n, LA, LB, first=0, device=dev)
    A = A.reshape(-1).contiguous()
    offA = torch.arange(0, (n + 1) * LA, LA, dtype=torch.int64, device=dev)
    os.environ["POLYHIP_SW_TILE64"] = "1"  # (the wider tile's workspace: more pad blocks in front of and behind the profile)
    work = torch.empty(align.sw_workspace_bytes(sc, n, LA, LB, True) + 4096, dtype=torch.uint8, device=dev)
    res = {}
    for tag in ("1", "0"):
        os.environ["POLYHIP_SW_TILE64"] = tag
        score = torch.zeros(n, dtype=torch.int64, device=dev)
        ea, eb, er = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3))
        ts = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            align.sw_batch_dev(sc, A, offA, LA, B, None, LB, score, ea, eb, er, work)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        res[tag] = (score, ea, eb, er, sorted(ts)[1], align.last_path())
    os.environ.pop("POLYHIP_SW_TILE64", None)
    x, y = res["1"], res["0"]
    same = all(bool(torch.equal(x[i], y[i])) for i in range(4))
    print(f"{n} x {LA} bp vs {LB}: score pass 64 rows per lane (path {x[5]}) {x[4]:.2f} ms = {n * LA * LB / x[4] / 1e9:.2f}e12 cells/s; "
          f"128/152 rows per lane (path {y[5]}) {y[4]:.2f} ms = {n * LA * LB / y[4] / 1e9:.2f}e12; every pair equal: {same}, errors {int((x[3] != 0).sum())}", flush=True)
