"""Randomised sweep over what round 4 added (run on the GPU box, not part of the suite): every host-pointer entry point on
a RANDOM device list over the visible GPUs (ids repeat: 1..7 workers sharing the box's GPU) against the plain one-device call
and -- sampled -- against the oracle; the one-call reads -> distance-matrix pipeline; K4 at random strides / start ranges
(line ownership); seqhash over random lengths around the chunk and 64-chunk edges."""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, '.')
import oracle as orc
from poly_amd import align, alphabet, devices, mash, matrix, primers, seqhash
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
dev = torch.device('cuda:0')
ab = alphabet.NewAlphabet(list("-ACGT"))
om = orc.SubstitutionMatrix("-ACGT", "-ACGT", orc.NUC_4_SCORES)
stats = {}


def pack(seqs):
    offs = np.zeros(len(seqs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(q) for q in seqs])
    return np.frombuffer(b"".join(seqs) + b"\0", np.uint8)[:-1].copy(), offs


def dna(rng, L):
    return bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8))


def ragged(rng, n, hi):
    lens = rng.integers(0, hi, n)
    if n > 3 and rng.random() < 0.5:
        lens[int(rng.integers(0, n))] = hi * int(rng.integers(5, 60))  # one item that is most of the batch
    return [dna(rng, int(L)) for L in lens]


it = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed0 + it)
    it += 1
    ids = [0] * int(rng.integers(1, 8))
    what = int(rng.integers(0, 7))
    if what == 0:      # K1
        k, s = int(rng.choice([4, 17, 21, 31])), int(rng.choice([2, 16, 200, 1000]))
        reads = ragged(rng, int(rng.integers(1, 200)), 4000)
        buf, offs = pack(reads)
        prior = rng.integers(0, 2**32, (len(reads), s), dtype=np.uint32)
        one = mash.sketch_batch_packed(buf, offs, k, s, out=prior.copy())
        with devices.devices(ids):
            got = mash.sketch_batch_packed(buf, offs, k, s, out=prior.copy())
        assert (got == one).all() and (got == orc.mash_sketch_batch(buf, offs, k, s, out=prior.copy())).all(), ("k1", it)
    elif what == 1:    # reads -> matrix in one call
        k, s = 21, int(rng.choice([16, 100, 300]))
        gen = [dna(rng, 2500) for _ in range(4)]
        reads = []
        for i in range(int(rng.integers(2, 70))):
            g = bytearray(gen[i % 4])
            for j in rng.integers(0, len(g), 30):
                g[int(j)] = int(rng.choice(list(b"ACGT")))
            reads.append(bytes(g) if rng.random() > 0.1 else dna(rng, int(rng.integers(0, 400))))
        buf, offs = pack(reads)
        want_sk = orc.mash_sketch_batch(buf, offs, k, s)
        with devices.devices(ids if rng.random() < 0.8 else []):
            sk, c, d = mash.sketch_distance_matrix_packed(buf, offs, k, s)
        assert (sk == want_sk).all(), ("pipe sk", it)
        wc, wd = mash.distance_matrix_packed(want_sk, want_sk)
        assert (c == wc).all() and (d == wd).all(), ("pipe", it)
        i, j = int(rng.integers(0, len(reads))), int(rng.integers(0, len(reads)))
        a, b = orc.Mash(k, s), orc.Mash(k, s)
        a.Sketches, b.Sketches = want_sk[i].copy(), want_sk[j].copy()
        assert d[i, j] == a.Distance(b), ("pipe oracle", it)
    elif what == 2:    # SW / NW, shared or per-pair reference
        sc = align.NewScoring(matrix.NewSubstitutionMatrix(ab, ab, matrix.NUC_4), int(rng.choice([-1, -2, -5])))
        n = int(rng.integers(1, 300))
        ref = dna(rng, int(rng.integers(1, 600)))
        reads = ragged(rng, n, 180)
        A, offA = pack(reads)
        if rng.random() < 0.5:
            B, offB = pack([ref])[0], None
            refs = [ref] * n
        else:
            refs = ragged(rng, n, 120)
            B, offB = pack(refs)
        one = align.sw_align_strings_packed(sc, A, offA, B, offB)
        onen = align.nw_align_packed(sc, A, offA, B, offB)
        with devices.devices(ids):
            got = align.sw_align_strings_packed(sc, A, offA, B, offB)
            got2 = align.sw_align_packed(sc, A, offA, B, offB)
            gotn = align.nw_align_packed(sc, A, offA, B, offB)
        for q in range(4):
            assert (got[q] == one[q]).all() and (got2[q] == one[q]).all(), ("sw", it, q)
        assert got[4] == one[4] and got[5] == one[5] and got2[4] == one[4] and got2[5] == one[5], ("sw strings", it)
        assert (gotn[0] == onen[0]).all() and gotn[2:] == onen[2:], ("nw", it)
        p = int(rng.integers(0, n))
        assert (int(got[0][p]), got[4][p].decode(), got[5][p].decode()) == orc.smith_waterman(reads[p], refs[p], om, sc.GapPenalty)[:3], ("sw oracle", it)
    elif what == 3:    # K4 host scan on a list + _dev at random start ranges / plane strides
        g = dna(rng, int(rng.integers(30, 6000)))
        Lmin = int(rng.integers(1, 25))
        Lmax = Lmin + int(rng.integers(0, 14))
        want = orc.santalucia_scan(g, Lmin, Lmax, 500e-9, 50e-3, 0.0)
        with devices.devices(ids):
            got = primers.SantaLuciaScan(g, Lmin, Lmax)
        for a, b in zip(got, want):
            assert np.array_equal(a, b, equal_nan=True), ("k4 host", it, Lmin, Lmax, len(g))
        gt = torch.from_numpy(np.frombuffer(g, np.uint8).copy()).to(dev)
        nsall = len(g) - Lmin + 1
        a0 = int(rng.integers(0, nsall))
        ns = int(rng.integers(1, nsall - a0 + 1))
        ld = ns + int(rng.integers(0, 20))
        pad = int(rng.integers(0, 16))
        nl = Lmax - Lmin + 1
        outs = [torch.full((nl * ld + pad,), 7.0, dtype=torch.float64, device=dev) for _ in range(3)]
        primers.santalucia_scan_dev(gt, len(g), a0, ns, Lmin, Lmax, 500e-9, 50e-3, 0.0, *[o[pad:] for o in outs], ld)
        torch.cuda.synchronize()
        for o, w in zip(outs, want):
            v = o[pad:].view(nl, ld).cpu().numpy()
            assert np.array_equal(v[:, :ns], w[:, a0:a0 + ns], equal_nan=True), ("k4 dev", it, a0, ns, ld, pad)
            assert (v[:, ns:] == 7.0).all() and (o[:pad] == 7.0).all(), ("k4 dev wrote outside", it)
    elif what == 4:    # primer batches
        seqs = ragged(rng, int(rng.integers(1, 400)), 60)
        seqs = [q if q else b"A" for q in seqs]
        buf, offs = pack(seqs)
        one = primers.santalucia_batch_packed(buf, offs, 500e-9, 50e-3, 0.0)
        with devices.devices(ids):
            got = primers.santalucia_batch_packed(buf, offs, 500e-9, 50e-3, 0.0)
            md = primers.marmurdoty_batch_packed(buf, offs)
        assert all((a == b).all() for a, b in zip(got, one)), ("tm batch", it)
        p = int(rng.integers(0, len(seqs)))
        assert (got[0][p], got[1][p], got[2][p]) == orc.santalucia(seqs[p], 500e-9, 50e-3, 0.0) and md[p] == orc.marmur_doty(seqs[p]), ("tm oracle", it)
    elif what == 5:    # rotation + seqhash, lengths around the edges the kernels care about
        edges = [0, 1, 63, 64, 65, 1023, 1024, 1025, 4096, 5000, 7000, 7200, 65535, 65536, 65537, 66000]
        seqs = [dna(rng, int(rng.choice(edges) if rng.random() < 0.3 else rng.integers(0, 3000))) for _ in range(int(rng.integers(1, 60)))]
        if rng.random() < 0.3:
            unit = dna(rng, int(rng.integers(1, 40)))
            seqs.append(unit * int(rng.integers(2, 200)))
        buf, offs = pack(seqs)
        circ, ds = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        with devices.devices(ids):
            rot, out = seqhash.least_rotation_batch_packed(buf, offs, True)
            hs, err = seqhash.seqhash_batch_packed(buf, offs, 0, circ, ds)
        for p, q in enumerate(seqs):
            assert int(rot[p]) == orc.booth_least_rotation(q), ("k5", it, len(q))
            assert out[int(offs[p]):int(offs[p + 1])].tobytes() == orc.rotate_sequence(q), ("k5 rot", it)
            assert hs[p] == orc.seqhash(q, "DNA", circ, ds), ("seqhash", it, len(q), circ, ds)
    else:              # distance matrix rows over devices
        s = int(rng.choice([8, 64, 200]))
        ny, nx = int(rng.integers(1, 300)), int(rng.integers(1, 120))
        fam = [np.sort(rng.integers(0, 1 << 24, s, dtype=np.uint32)) for _ in range(5)]
        mk = lambda: np.sort(np.where(rng.random(s) < 0.2, rng.integers(0, 1 << 24, s, dtype=np.uint32), fam[int(rng.integers(0, 5))]))
        Y = np.stack([mk() for _ in range(ny)])
        X = np.stack([mk() for _ in range(nx)])
        with devices.devices(ids):
            c, d = mash.distance_matrix_packed(X, Y)
        assert (d == orc.mash_distance_matrix(X, Y)).all(), ("k2", it)
    stats[what] = stats.get(what, 0) + 1
print(f"fuzz_r04: {it} iterations in {budget:.0f} s, seed {seed0}: all equal; per case {dict(sorted(stats.items()))}")
