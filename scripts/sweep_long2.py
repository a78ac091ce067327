"""Parity sweep of the Smith-Waterman / Needleman-Wunsch paths that round 5 left on samples (round 6, verdict item 5b),
every pair of every leg against oracle/poly_oracle.c on the host cores -- score, end cell and BOTH aligned strings:

  2 kb    5,000 reads of 1025..2048 bp vs one 5 kb reference    (packed multi-lane score pass / one-wave-per-pair kernels, table form)
  4 kb    5,000 reads of 2049..4096 bp vs one 5 kb reference
  pairs   5,000 pairs of up to 600 x 600 bp, per-pair B          (one wave per pair)
  nw      5,000 pairs of up to 1000 x 1000 bp, NeedlemanWunsch   (nw_wave_kernel; align.go:100-166 incl. its either-index-0 stop)

Reads are windows of their reference with 6 % substitutions and 2 % indels, ragged lengths, the first one at the maximum.

    python scripts/sweep_long2.py [pairs per leg] > profiles/r06_sweep_long2.log
"""
import concurrent.futures as cf
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as orc  # noqa: E402
from poly_amd import align, alphabet, matrix  # noqa: E402

npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ab = alphabet.NewAlphabet(list("-ACGT"))
sc = align.NewScoring(matrix.NewSubstitutionMatrix(ab, ab, matrix.NUC_4), -2)
om = orc.SubstitutionMatrix("-ACGT", "-ACGT", orc.NUC_4_SCORES)
ncpu = max(1, min(os.cpu_count() or 1, 64))
ACGT = np.frombuffer(b"ACGT", np.uint8)


def mutate(rng, seq: bytes, sub=0.06, indel=0.02) -> bytes:
    a = np.frombuffer(seq, np.uint8).copy()
    hit = rng.random(len(a)) < sub
    a[hit] = ACGT[rng.integers(0, 4, int(hit.sum()))]
    out = bytearray()
    r = rng.random(len(a))
    for i, c in enumerate(a):
        if r[i] < indel / 2:
            continue                      # deletion
        out.append(int(c))
        if r[i] > 1 - indel / 2:
            out.append(int(ACGT[rng.integers(0, 4)]))  # insertion
    return bytes(out)


def pack(seqs):
    offs = np.zeros(len(seqs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(x) for x in seqs])
    return np.frombuffer(b"".join(seqs), np.uint8).copy(), offs


def b(x):
    return x if isinstance(x, bytes) else x.encode("latin-1")


worst = 0
for leg, lo, hi, shared, nw in (("2 kb", 1025, 2048, True, False), ("4 kb", 2049, 4096, True, False),
                                ("per-pair B 600 x 600", 300, 600, False, False), ("NeedlemanWunsch 1000 x 1000", 500, 1000, False, True)):
    rng = np.random.default_rng(0x10A7 + hi)
    ref = orc.synth_dna(0xC4, 5000).tobytes()
    reads, refs = [], []
    for p in range(npairs):
        L = int(rng.integers(lo, hi + 1)) if p else hi
        if shared:
            at = int(rng.integers(0, 5000 - L + 1))
            reads.append(mutate(rng, ref[at:at + L])[:hi])
            refs.append(ref)
        else:
            LBp = int(rng.integers(lo, hi + 1)) if p != 1 else hi
            r = orc.synth_dna(0x5000 + p, LBp).tobytes()
            # the read: a mutated window of ITS reference, wrapped so that it may be longer than what the reference offers
            at = int(rng.integers(0, LBp))
            reads.append(mutate(rng, (r * 3)[at:at + L])[:hi])
            refs.append(r)
    A, offA = pack(reads)
    t0 = time.time()
    if nw:
        B, offB = pack(refs)
        score, err, sa, sb = align.nw_align_packed(sc, A, offA, B, offB)
        paths = (align.nw_last_path(),)
        got = [(int(score[p]), sa[p], sb[p]) for p in range(npairs)]
    else:
        B, offB = (np.frombuffer(ref, np.uint8).copy(), None) if shared else pack(refs)
        score, ea, eb, err, sa, sb = align.sw_align_packed(sc, A, offA, B, offB)
        paths = (align.last_path(), align.sw_traceback_last_path())
        got = [(int(score[p]), int(ea[p]), int(eb[p]), sa[p], sb[p]) for p in range(npairs)]
    t_gpu = time.time() - t0
    assert int(np.abs(err).sum()) == 0

    def one(p):
        if nw:
            s, wa, wb = orc.needleman_wunsch(reads[p], refs[p], om, -2)
            want = (s, b(wa), b(wb))
        else:
            s, wa, wb, wea, web = orc.smith_waterman(reads[p], refs[p], om, -2)
            want = (s, wea, web, b(wa), b(wb))
        return None if got[p] == want else f"pair {p}: got {got[p][:3]} want {want[:3]}"

    t0 = time.time()
    with cf.ThreadPoolExecutor(ncpu) as ex:
        bad = [x for x in ex.map(one, range(npairs)) if x]
    dt = time.time() - t0
    lens = np.diff(offA.astype(np.int64))
    print(f"{leg}: {npairs} pairs (rows {int(lens.min())}..{int(lens.max())}), paths {paths}, host call {t_gpu:.1f} s; EVERY pair -- score"
          f"{'' if nw else ' + endA + endB'} + both aligned strings -- vs the oracle on {ncpu} threads in {dt:.1f} s: {len(bad)} differ "
          f"(scores {int(np.min(score))}..{int(np.max(score))})", flush=True)
    for x in bad[:5]:
        print(x)
    worst = max(worst, len(bad))
sys.exit(1 if worst else 0)
