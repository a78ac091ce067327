"""long reads against one reference, score + strings in ONE call (polyhip_sw_align_batch_dev): the end cell left to the
traceback kernel (default) against POLYHIP_SW_FUSE=0 (locate in the score pass) -- every output of every pair compared"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poly_amd import align, alphabet, matrix, workloads

dev = torch.device("cuda:0")
a = alphabet.NewAlphabet(list("-ACGT"))
sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
for n, LA, LB in ((80_000, 1000, 5000), (160_000, 500, 5000)):
    B, A = workloads.config4_reads(n, LA, LB, first=0, device=dev)
    A = A.reshape(-1).contiguous()
    offA = torch.arange(0, (n + 1) * LA, LA, dtype=torch.int64, device=dev)
    work = torch.empty(align.sw_workspace_bytes(sc, n, LA, LB, True), dtype=torch.uint8, device=dev)
    stride = align.sw_traceback_stride(sc, LA, LB)
    tbw = torch.empty(align.sw_traceback_workspace_bytes(sc, n, LA, LB), dtype=torch.uint8, device=dev)
    res = {}
    for tag, env in (("end cell in the traceback", None), ("locate in the score pass", "0")):
        if env is None:
            os.environ.pop("POLYHIP_SW_FUSE", None)
        else:
            os.environ["POLYHIP_SW_FUSE"] = env
        score = torch.zeros(n, dtype=torch.int64, device=dev)
        ea, eb, er, ln = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4))
        alnA = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
        alnB = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            align.sw_align_dev(sc, A, offA, LA, B, None, LB, score, ea, eb, er, alnA, alnB, ln, work, tbw)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        res[tag] = (score, ea, eb, er, ln, alnA, alnB, sorted(ts)[1], (align.last_path(), align.sw_traceback_last_path()))
    os.environ.pop("POLYHIP_SW_FUSE", None)
    x, y = res["end cell in the traceback"], res["locate in the score pass"]
    live = torch.arange(stride, device=dev)[None, :] >= (stride - x[4].long())[:, None]
    same = all(bool(torch.equal(x[i], y[i])) for i in range(5)) and bool(((x[5] == y[5]) | ~live).all()) and bool(((x[6] == y[6]) | ~live).all())
    print(f"{n} x {LA} bp vs {LB}, one call (paths {x[8]}): end cell in the traceback {x[7]:.2f} ms = {n * LA * LB / x[7] / 1e9:.2f}e12 cells/s, "
          f"locate in the score pass {y[7]:.2f} ms; every pair equal: {same}", flush=True)
