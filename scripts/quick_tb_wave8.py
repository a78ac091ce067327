"""one-wave-per-pair traceback (257..1024 rows): the byte-profile sweep (path 7) against its table form (POLYHIP_TB_WAVE8=0,
path 4) on bench_extra's long-read legs -- every pair's strings equal, the two times side by side."""
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
    score = torch.zeros(n, dtype=torch.int64, device=dev)
    ea, eb, er = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3))
    work = torch.empty(align.sw_workspace_bytes(sc, n, LA, LB, True), dtype=torch.uint8, device=dev)
    align.sw_batch_dev(sc, A, offA, LA, B, None, LB, score, ea, eb, er, work)
    stride = align.sw_traceback_stride(sc, LA, LB)
    tbw = torch.empty(align.sw_traceback_workspace_bytes(sc, n, LA, LB), dtype=torch.uint8, device=dev)
    res = {}
    for tag, env in (("byte profile", {}), ("table", {"POLYHIP_TB_WAVE8": "0"}), ("plain walk", {"POLYHIP_TB_WALKREG": "0"}),
                     ("no walk", {"POLYHIP_TB_NOWALK": "1"})):
        for k in ("POLYHIP_TB_WAVE8", "POLYHIP_TB_WALKREG", "POLYHIP_TB_NOWALK"):
            os.environ.pop(k, None)
        os.environ.update(env)
        alnA = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
        alnB = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
        ln = torch.zeros(n, dtype=torch.int32, device=dev)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            align.sw_traceback_dev(sc, A, offA, LA, B, None, LB, ea, eb, er, alnA, alnB, ln, tbw, score_t=score)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        res[tag] = (alnA, alnB, ln, sorted(ts)[1], align.sw_traceback_last_path())
    for k in ("POLYHIP_TB_WAVE8", "POLYHIP_TB_WALKREG", "POLYHIP_TB_NOWALK"):
        os.environ.pop(k, None)
    z = res["plain walk"]
    print(f"   byte profile + plain walk {z[3]:.2f} ms (equal: {bool(torch.equal(z[2], res['byte profile'][2]) and torch.equal(z[0], res['byte profile'][0]) and torch.equal(z[1], res['byte profile'][1]))}), "
          f"sweep alone {res['no walk'][3]:.2f} ms", flush=True)
    x, y = res["byte profile"], res["table"]
    live = torch.arange(stride, device=dev)[None, :] >= (stride - x[2].long())[:, None]
    same = bool(torch.equal(x[2], y[2]) and bool(((x[0] == y[0]) | ~live).all()) and bool(((x[1] == y[1]) | ~live).all()))
    print(f"{n} x {LA} bp vs {LB}: byte profile (path {x[4]}) {x[3]:.2f} ms, table (path {y[4]}) {y[3]:.2f} ms, every pair equal: {same}, "
          f"mean length {float(x[2].double().mean()):.1f}", flush=True)
