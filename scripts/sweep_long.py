"""Parity sweep of the long-read Smith-Waterman paths against the oracle at bench size (round 5): bench_extra's 1 kb and 500 bp
legs -- reads that are windows of the 5 kb reference with 5 % substitutions and 1 % indels (poly_amd.workloads.config4_reads)
-- through polyhip_sw_align_batch_dev (packed multi-lane score pass, end cells left to the byte-profile one-wave-per-pair
traceback): score, endA, endB and BOTH aligned strings of `pairs` sampled pairs per leg (incl. the first and the last)
against oracle/poly_oracle.c orc_smith_waterman (align.go:171-232) on the host cores.

    python scripts/sweep_long.py [pairs]  > profiles/r05_sweep_long.log
"""
import concurrent.futures as cf
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as orc  # noqa: E402
from poly_amd import align, alphabet, matrix, workloads  # noqa: E402

npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
dev = torch.device("cuda:0")
ab = alphabet.NewAlphabet(list("-ACGT"))
sc = align.NewScoring(matrix.NewSubstitutionMatrix(ab, ab, matrix.NUC_4), -2)
om = orc.SubstitutionMatrix("-ACGT", "-ACGT", orc.NUC_4_SCORES)
ncpu = max(1, min(os.cpu_count() or 1, 128))
worst = 0
for n, LA, LB in ((80_000, 1000, 5000), (160_000, 500, 5000)):
    B, A2 = workloads.config4_reads(n, LA, LB, device=dev)
    A = A2.reshape(-1).contiguous()
    offA = torch.arange(0, (n + 1) * LA, LA, dtype=torch.int64, device=dev)
    stride = align.sw_traceback_stride(sc, LA, LB)
    score = torch.zeros(n, dtype=torch.int64, device=dev)
    ea, eb, er, ln = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4))
    alnA = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    alnB = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    work = torch.empty(align.sw_workspace_bytes(sc, n, LA, LB, True), dtype=torch.uint8, device=dev)
    tbw = torch.empty(align.sw_traceback_workspace_bytes(sc, n, LA, LB), dtype=torch.uint8, device=dev)
    align.sw_align_dev(sc, A, offA, LA, B, None, LB, score, ea, eb, er, alnA, alnB, ln, work, tbw)
    torch.cuda.synchronize()
    paths = (align.last_path(), align.sw_traceback_last_path())
    assert int(er.abs().sum()) == 0
    rng = np.random.default_rng(0x10A6 + LA)
    m = min(npairs, n)
    sample = np.sort(np.concatenate([rng.choice(n - 2, m - 2, replace=False) + 1, [0, n - 1]]))
    idx = torch.from_numpy(sample).to(dev)
    h = {k: v[idx].cpu().numpy() for k, v in dict(score=score, ea=ea, eb=eb, ln=ln, A=A2, alnA=alnA, alnB=alnB).items()}
    refb = B.cpu().numpy().tobytes()

    def one(j):
        ws, wa, wb, wea, web = orc.smith_waterman(h["A"][j].tobytes(), refb, om, -2)
        wa = wa if isinstance(wa, bytes) else wa.encode("latin-1")
        wb = wb if isinstance(wb, bytes) else wb.encode("latin-1")
        L = int(h["ln"][j])
        got = (int(h["score"][j]), int(h["ea"][j]), int(h["eb"][j]), h["alnA"][j, stride - L:].tobytes(), h["alnB"][j, stride - L:].tobytes())
        return None if got == (ws, wea, web, wa, wb) else f"pair {sample[j]}: got {got[:3]} want {(ws, wea, web)}"

    t0 = time.time()
    with cf.ThreadPoolExecutor(ncpu) as ex:
        bad = [b for b in ex.map(one, range(len(sample))) if b]
    dt = time.time() - t0
    print(f"{n} x {LA} bp vs {LB} (5 % subs + 1 % indels), one call, paths {paths}: {len(sample)} pairs incl. the first and the last, score + "
          f"endA + endB + both aligned strings vs orc_smith_waterman on {ncpu} threads in {dt:.1f} s: {len(bad)} differ "
          f"(scores {int(h['score'].min())}..{int(h['score'].max())}, aligned lengths {int(h['ln'].min())}..{int(h['ln'].max())})", flush=True)
    for b in bad[:5]:
        print(b)
    worst = max(worst, len(bad))
    del alnA, alnB, work, tbw
    torch.cuda.empty_cache()
sys.exit(1 if worst else 0)
