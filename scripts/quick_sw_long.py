"""long reads against one reference (path 7): score pass with the locate step on a byte profile (default) against the general
one-wave-per-pair kernel in locate mode (POLYHIP_SW_WAVE8=0) -- every pair's four outputs equal, the two times side by side"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poly_amd import align, alphabet, matrix, workloads

dev = torch.device("cuda:0")
a = alphabet.NewAlphabet(list("-ACGT"))
sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
for n, LA, LB in ((80_000, 1000, 5000), (160_000, 500, 5000), (40_000, 700, 3000)):
    B, A = workloads.config4_reads(n, LA, LB, first=0, device=dev)
    A = A.reshape(-1).contiguous()
    offA = torch.arange(0, (n + 1) * LA, LA, dtype=torch.int64, device=dev)
    work = torch.empty(align.sw_workspace_bytes(sc, n, LA, LB, True), dtype=torch.uint8, device=dev)
    res = {}
    for tag, env in (("byte profile", None), ("general", "0")):
        if env is None:
            os.environ.pop("POLYHIP_SW_WAVE8", None)
        else:
            os.environ["POLYHIP_SW_WAVE8"] = env
        score = torch.zeros(n, dtype=torch.int64, device=dev)
        ea, eb, er = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3))
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            align.sw_batch_dev(sc, A, offA, LA, B, None, LB, score, ea, eb, er, work)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        res[tag] = (score, ea, eb, er, sorted(ts)[1], align.last_path())
    os.environ.pop("POLYHIP_SW_WAVE8", None)
    x, y = res["byte profile"], res["general"]
    same = all(bool(torch.equal(x[i], y[i])) for i in range(4))
    print(f"{n} x {LA} bp vs {LB}: score pass (path {x[5]}) locate on a byte profile {x[4]:.2f} ms = {n * LA * LB / x[4] / 1e9:.2f}e12 cells/s, "
          f"general kernel {y[4]:.2f} ms, every pair equal: {same}, errors {int((x[3] != 0).sum())}", flush=True)
