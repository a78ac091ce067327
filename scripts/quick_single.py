"""Latency of single calls through the Python mirror (host-pointer C ABI: alloc + copy + kernel + copy)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from poly_amd import mash, align, alphabet, matrix, primers, seqhash
phix = open('tests/golden/phix174.seq').read().strip()
puc = open('tests/golden/puc19.seq').read().strip()
a = alphabet.NewAlphabet(list("-ACGT"))
sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
read = phix[1000:1150].upper()
ref = phix[:5000].upper()
def t(name, f, reps=20):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    print(f"{name:48s} {(time.perf_counter()-t0)/reps*1e3:8.3f} ms")
m = mash.New(21, 1000)
t("mash.Sketch(phiX174, k=21, s=1000)", lambda: m.Sketch(phix))
m2 = mash.New(21, 1000); m2.Sketch(phix[::-1])
t("mash.Distance", lambda: m.Distance(m2))
t("align.SmithWaterman(150 bp, 5000 bp)", lambda: align.SmithWaterman(read, ref, sc))
t("align.NeedlemanWunsch(150 bp, 150 bp)", lambda: align.NeedlemanWunsch(read, ref[1000:1150], sc))
t("primers.MeltingTemp(20-mer)", lambda: primers.MeltingTemp("GTAAAACGACGGCCAGTACG"))
t("seqhash.RotateSequence(pUC19)", lambda: seqhash.RotateSequence(puc))
t("seqhash.Hash(pUC19, DNA, circular, ds)", lambda: seqhash.Hash(puc, "DNA", True, True))
