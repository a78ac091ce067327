"""Randomised sweep over the kernels fuzz_k1 / fuzz_k3 do not reach (run on the GPU box, not part of the suite):
NeedlemanWunsch of every length class, long-read SmithWaterman, shared counts on random family structures
(sparse, dense rows, repeated hashes, irregular sketches), least rotation + seqhash on low-complexity input."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import oracle as orc
from poly_amd import align, alphabet, matrix, mash, seqhash
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
it = 0
def pack(seqs):
    offs = np.zeros(len(seqs) + 1, np.uint64); offs[1:] = np.cumsum([len(q) for q in seqs])
    return np.frombuffer(b"".join(seqs) + b"\0", np.uint8)[:-1].copy(), offs
def b2(x): return x if isinstance(x, bytes) else x.encode()
while time.time() < t_end:
    rng = np.random.default_rng(seed0 + it); it += 1
    syms = "ACGT"; a = alphabet.NewAlphabet(list(syms))
    mat = rng.integers(-5, 6, (4, 4)).astype(int); np.fill_diagonal(mat, rng.integers(1, 8, 4)); mat = mat.tolist()
    gap = -int(rng.integers(1, 6))
    sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, mat), gap); om = orc.SubstitutionMatrix(syms, syms, mat)
    # --- NW + long SW: one length class per iteration
    maxlen = int(rng.choice([40, 150, 250, 400, 900, 1800, 3500]))
    A, B = [], []
    for _ in range(6):
        la = int(rng.integers(1, maxlen + 1)); x = bytes(rng.choice(list(b"ACGT"), la).astype(np.uint8))
        y = bytearray(x)
        for _ in range(la // 15 + 1):
            if y and rng.random() < 0.4: del y[int(rng.integers(0, len(y)))]
            else: y.insert(int(rng.integers(0, len(y) + 1)), int(rng.choice(list(b"ACGT"))))
        A.append(x); B.append(bytes(y))
    A[0] = bytes(rng.choice(list(b"ACGT"), maxlen).astype(np.uint8))
    pa, oa = pack(A); pb, ob = pack(B)
    score, err, sa, sb = align.nw_align_packed(sc, pa, oa, pb, ob)
    for p in range(len(A)):
        w = orc.needleman_wunsch(A[p], B[p], om, gap)
        assert (int(score[p]), sa[p], sb[p]) == (w[0], b2(w[1]), b2(w[2])), ("nw", it, p, maxlen, gap)
    got = align.sw_align_packed(sc, pa, oa, pb, ob)
    for p in range(len(A)):
        s, xa, xb, ea, eb = orc.smith_waterman(A[p], B[p], om, gap)
        assert (int(got[0][p]), int(got[1][p]), int(got[2][p]), got[4][p], got[5][p]) == (s, ea, eb, b2(xa), b2(xb)), ("sw", it, p, maxlen, gap)
    # --- shared counts on a random family structure
    s_ = int(rng.choice([8, 32, 100])); ny = int(rng.choice([50, 400, 3000]))
    Y = np.sort(rng.integers(0, 1 << 30, (ny, s_), dtype=np.uint32), axis=1)
    for _ in range(int(rng.integers(1, 5))):
        fam = rng.choice(ny, int(rng.integers(2, min(ny, 2500) + 1)), replace=False)
        share = int(rng.integers(1, s_ + 1)); vals = np.sort(rng.integers(0, 1 << 30, share, dtype=np.uint32))
        if rng.random() < 0.3: vals[:] = vals[0]                     # one repeated hash
        Y[fam, :share] = vals
    Y = np.sort(Y, axis=1)
    for q in rng.choice(ny, int(rng.integers(0, 3)), replace=False): Y[q] = Y[q][::-1]   # irregular
    X = Y[rng.choice(ny, min(ny, 12), replace=False)]
    cnt, dist = mash.distance_matrix_packed(X, Y)
    for i in range(len(X)):
        for j in rng.choice(ny, min(ny, 40), replace=False):
            assert int(cnt[i, j]) == orc.mash_shared(X[i], Y[j]), ("k2", it, i, int(j), s_, ny)
    # --- least rotation + seqhash on low-complexity / periodic input
    seqs = []
    for _ in range(12):
        n = int(rng.integers(1, 4000)); kind = int(rng.integers(0, 4))
        if kind == 0: q = bytes(rng.choice(list(b"ACGT"), n).astype(np.uint8))
        elif kind == 1: u = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 60))).astype(np.uint8)); q = (u * (n // len(u) + 1))[:n]
        elif kind == 2:
            q = bytearray(rng.choice(list(b"ACGT"), n).astype(np.uint8)); r0 = int(rng.integers(0, n)); rl = int(rng.integers(1, n + 1))
            for t in range(rl): q[(r0 + t) % n] = ord("A")
            q = bytes(q)
        else: q = bytes([int(rng.choice(list(b"ACGT")))]) * n
        seqs.append(q)
    buf, offs = pack(seqs)
    rot, out = seqhash.least_rotation_batch_packed(buf, offs, True)
    for i, q in enumerate(seqs):
        assert int(rot[i]) == orc.booth_least_rotation(q) and out[int(offs[i]):int(offs[i + 1])].tobytes() == orc.rotate_sequence(q), ("k5", it, i, len(q))
    hs = seqhash.HashBatch([q.decode() for q in seqs], "DNA", True, True)
    for i, q in enumerate(seqs):
        assert hs[i] == orc.seqhash(q, "DNA", True, True), ("seqhash", it, i)
    if it % 10 == 0: print(f"it {it}: maxlen {maxlen} gap {gap} s {s_} ny {ny} ok", flush=True)
print("fuzz done", it, "iterations")
