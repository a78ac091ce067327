"""How far apart are the 4-column blocks that hold a Smith-Waterman maximum?  (round 6: the reason the locate step can resolve
ties itself.)  Plain numpy DP of configs[3]'s reads (poly_amd.workloads.config4_reads: windows of the 5 kb reference with 5 %
substitutions + 1 % indels; NUC_4, gap -2) -- every cell worth the pair's maximum, the blocks they lie in.  No GPU, no oracle.

    python scripts/tie_stats.py [pairs] > profiles/r06_tie_stats.log      (30,000 pairs: ~20 min on one core)
"""
import sys
import numpy as np
sys.path.insert(0, '.')
from poly_amd import workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
ref, reads = workloads.config4_reads(n)
ref, reads = np.asarray(ref), np.asarray(reads)
LB, g = len(ref), -2
jj = np.arange(LB + 1)
spans = []
for p in range(n):
    a = reads[p]
    Hprev = np.zeros(LB + 1, np.int32)
    rows = []
    for i in range(len(a)):
        s = np.where(ref == a[i], 5, -4).astype(np.int32)
        T = np.zeros(LB + 1, np.int32)
        T[1:] = np.maximum(0, np.maximum(Hprev[:-1] + s, Hprev[1:] + g))
        # H[j] = max(T[j], H[j-1] + g) = g*j + running max of (T[k] - g*k): the row's left dependency as a prefix maximum
        H = np.maximum(np.maximum.accumulate(T - g * jj) + g * jj, 0)
        rows.append(H)
        Hprev = H
    Hm = np.stack(rows)
    M = Hm.max()
    _, cc = np.nonzero(Hm == M)
    blocks = np.unique((cc - 1) // 4)
    if len(blocks) > 1:
        spans.append((int(blocks[-1] - blocks[0]), len(blocks), int(M)))
d = np.array([x[0] for x in spans]) if spans else np.zeros(0, int)
print(f"{n} pairs of configs[3]: {len(spans)} ({100 * len(spans) / n:.2f} %) have their maximum in more than one 4-column block; "
      f"last - first block: <= 1: {(d <= 1).sum()}, <= 3: {(d <= 3).sum()}, <= 15: {(d <= 15).sum()}, > 15: {(d > 15).sum()}")
print("(span, blocks, M) of the first 20:", sorted(spans)[:20])
