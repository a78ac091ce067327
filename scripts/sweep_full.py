"""Full-size parity sweep against the oracle, beyond what the -m gpu tests afford (round-4 verdict, item 2c):
BASELINE configs[3] -- 100,000 of the contract's 1,000,000 pairs (poly_amd.workloads.config4_reads: 150 bp windows of the 5 kb
reference with 5 % substitutions and 1 % indels), score, endA, endB and BOTH aligned strings from polyhip_sw_align_batch_dev
against oracle/poly_oracle.c orc_smith_waterman (align.go:171-232) on every host core.  ~3 min on 16 cores.

    python scripts/sweep_full.py [pairs]  > profiles/r05_sweep_full.log
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

npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dev = torch.device("cuda:0")
n, LA, LB = 1_000_000, 150, 5000
B, A2 = workloads.config4_reads(n, LA, LB, device=dev)
A = A2.reshape(-1).contiguous()
offA = torch.arange(0, (n + 1) * LA, LA, dtype=torch.int64, device=dev)
ab = alphabet.NewAlphabet(list("-ACGT"))
sc = align.NewScoring(matrix.NewSubstitutionMatrix(ab, ab, matrix.NUC_4), -2)
om = orc.SubstitutionMatrix("-ACGT", "-ACGT", orc.NUC_4_SCORES)
stride = align.sw_traceback_stride(sc, LA, LB)
score = torch.zeros(n, dtype=torch.int64, device=dev)
ea, eb, er, ln = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4))
alnA = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
alnB = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
work = torch.empty(align.sw_workspace_bytes(sc, n, LA, LB, True), dtype=torch.uint8, device=dev)
tbw = torch.empty(align.sw_traceback_workspace_bytes(sc, n, LA, LB), dtype=torch.uint8, device=dev)
align.sw_align_dev(sc, A, offA, LA, B, None, LB, score, ea, eb, er, alnA, alnB, ln, work, tbw)
torch.cuda.synchronize()
assert int(er.abs().sum()) == 0
rng = np.random.default_rng(0x5EE9)
sample = np.sort(np.concatenate([rng.choice(n - 2, npairs - 2, replace=False) + 1, [0, n - 1]]))
idx = torch.from_numpy(sample).to(dev)
h = {k: v[idx].cpu().numpy() for k, v in dict(score=score, ea=ea, eb=eb, ln=ln, A=A2, alnA=alnA, alnB=alnB).items()}
refb = B.cpu().numpy().tobytes()
ncpu = max(1, min(os.cpu_count() or 1, 64))


def one(j):
    ws, wa, wb, wea, web = orc.smith_waterman(h["A"][j].tobytes(), refb, om, -2)
    wa = wa if isinstance(wa, bytes) else wa.encode("latin-1")
    wb = wb if isinstance(wb, bytes) else wb.encode("latin-1")
    L = int(h["ln"][j])
    got = (int(h["score"][j]), int(h["ea"][j]), int(h["eb"][j]), h["alnA"][j, stride - L:].tobytes(), h["alnB"][j, stride - L:].tobytes())
    return None if got == (ws, wea, web, wa, wb) else f"pair {sample[j]}: got {got} want {(ws, wea, web, wa, wb)}"


t0 = time.time()
with cf.ThreadPoolExecutor(ncpu) as ex:
    bad = [b for b in ex.map(one, range(len(sample))) if b]
dt = time.time() - t0
print(f"configs[3] (1M x 150 bp vs 5 kb, 5 % subs + 1 % indels): {len(sample)} pairs incl. the first and the last, score + endA + endB + "
      f"both aligned strings vs orc_smith_waterman on {ncpu} threads in {dt:.1f} s: {len(bad)} differ"
      f" (scores {int(h['score'].min())}..{int(h['score'].max())}, aligned lengths {int(h['ln'].min())}..{int(h['ln'].max())})")
for b in bad[:5]:
    print(b)
sys.exit(1 if bad else 0)
