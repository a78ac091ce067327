"""Host-side mirror of the reference interface (no GPU): alphabet, substitution matrix, Scoring
flattening, packing, pcr's reverse complement, error classes.

Mirrors alphabet/alphabet_test.go:10-73, search/align/matrix/matrix_test.go:11-49 and the Scoring
conventions of search/align/align.go:73-95."""
import numpy as np
import pytest

import oracle as orc
from poly_amd import _lib, align, alphabet, matrix, pcr
from poly_amd.mash import _pack


def test_alphabet():
    """alphabet/alphabet_test.go:10-73"""
    symbols = ["A", "C", "G", "T"]
    a = alphabet.NewAlphabet(symbols)
    for i, s in enumerate(symbols):
        assert a.Encode(s) == i and a.Decode(i) == s
    with pytest.raises(alphabet.Error, match="Symbol X not in alphabet"):
        a.Encode("X")
    with pytest.raises(alphabet.Error):
        a.Decode(len(symbols))
    ext = a.Extend(["N", "-", "*"])
    assert [ext.Encode(s) for s in symbols] == [0, 1, 2, 3]
    assert [ext.Encode(s) for s in ["N", "-", "*"]] == [4, 5, 6]
    assert a.Symbols() == symbols
    assert alphabet.DNA.Symbols() == ["A", "C", "G", "T"] and alphabet.RNA.Symbols()[-1] == "U"
    # a repeated symbol keeps its last index (alphabet.go:27-30)
    assert alphabet.NewAlphabet(["A", "C", "A"]).Encode("A") == 2


def test_substitution_matrix():
    """search/align/matrix/matrix_test.go:11-49"""
    a1 = alphabet.NewAlphabet(["-", "A", "C", "G", "T"])
    m = matrix.NewSubstitutionMatrix(a1, alphabet.NewAlphabet(["-", "A", "C", "G", "T"]), matrix.NUC_4)
    assert [m.Score(x, y) for x, y in (("A", "A"), ("A", "C"), ("C", "T"), ("-", "-"))] == [5, -4, -4, 0]
    with pytest.raises(alphabet.Error):
        m.Score("X", "A")
    with pytest.raises(ValueError):
        matrix.NewSubstitutionMatrix(a1, a1, [[0, 1], [1, 0]])
    assert matrix.Default.Score("Q", "Q") == 1 and matrix.Default.Score("Q", "R") == -1  # matrix.go:40-73


def test_scoring_flatten_matches_the_oracle_flatten():
    """the Go wrapper can only see the matrix through Score(): the 256x256 table + valid masks it builds
    must be what the oracle's restatement of matrix.go:28-38 gives, also for asymmetric two-alphabet matrices"""
    rows, cols = "ACGT", "ACGTN"
    scores = [[4, -2, -1, -3, 0], [-2, 5, -3, -1, 0], [-1, -4, 6, -2, 0], [-3, -1, -2, 3, 0]]
    sc = align.NewScoring(matrix.NewSubstitutionMatrix(alphabet.NewAlphabet(list(rows)), alphabet.NewAlphabet(list(cols)),
                                                       scores), -3)
    lut, va, vb = sc.flatten()
    om = orc.SubstitutionMatrix(rows, cols, scores)
    for a in range(256):
        for b in range(0, 256, 3):
            ok_a, ok_b = a < 128 and chr(a) in rows, b < 128 and chr(b) in cols
            assert va[a] == ok_a and vb[b] == ok_b
            if ok_a and ok_b:
                assert lut[a, b] == scores[rows.index(chr(a))][cols.index(chr(b))]
    assert sc.Score(ord("G"), ord("N")) == 0 and sc.GapPenalty == -3
    assert align.NewScoring(None, -1).SubstitutionMatrix is matrix.Default  # align.go:80-82
    _ = om


def test_pack_and_revcomp():
    buf, offs = _pack(["ACGT", b"", "TT", b"\xff\x00"])
    assert buf.tobytes() == b"ACGTTT\xff\x00" and offs.tolist() == [0, 4, 4, 6, 8]
    buf, offs = _pack([])
    assert len(buf) == 0 and offs.tolist() == [0]
    for s in (b"GATTACA", b"acgtNNRYKM", b"AU-*", b""):
        assert pcr._revcomp(s) == orc.reverse_complement(s)  # transform.go:15-23,78-109


def test_error_classes_and_status_codes():
    assert issubclass(_lib.GoPanic, _lib.PolyhipError)
    e = _lib.PolyhipError(-2, "x")
    assert e.status == -2 and "x" in str(e)
    # the binding table and the header agree (names + arity are checked against the library in test_abi_cpu)
    assert "polyhip_fasta_pack_dev" in _lib.SIGNATURES and len(_lib.SIGNATURES) >= 40


def test_synthetic_workloads_are_pinned():
    """poly_amd/workloads.py (SURVEY 8d inputs) is the one definition bench.py, the GPU tests and the CPU baselines draw
    from: the DNA stream equals the oracle's, slices equal the whole, and the config-4 read generator (5 % substitutions +
    1 % indels, integer torch ops only) is pinned on a digest so that a change of its bytes cannot go unnoticed"""
    import hashlib
    from poly_amd import workloads as w
    assert (w.synth_dna(0xC2, 5000) == orc.synth_dna(0xC2, 5000)).all()
    assert (w.synth_dna(0xC2, 777, first=4321) == orc.synth_dna(0xC2, 6000)[4321:4321 + 777]).all()
    ref, reads = w.config4_reads(1000)
    assert (ref == orc.synth_dna(0xC4, 5000)).all() and reads.shape == (1000, 150)
    assert hashlib.sha256(reads.tobytes()).hexdigest() == "1f08f2aa6b80dbd7aa473bdef81dac58de9e666aa7981d0dbdaf3773e023f598"
    assert (w.config4_reads(7, first=500)[1] == reads[500:507]).all()
    # the channel does what it says: ~5 % of the aligned columns mismatch, ~1 % are gaps
    om = orc.SubstitutionMatrix("-ACGT", "-ACGT", orc.NUC_4_SCORES)
    mism = gaps = cols = 0
    for r in reads[:60]:
        _, a, b, _, _ = orc.smith_waterman(r.tobytes(), ref.tobytes(), om, -2)
        cols += len(a)
        gaps += a.count("-") + b.count("-")
        mism += sum(1 for x, y in zip(a, b) if x != y and x != "-" and y != "-")
    assert 0.025 < mism / cols < 0.08 and 0.003 < gaps / cols < 0.03


def test_bench_refuses_a_rank_count_it_was_not_started_with():
    """bench.py --gpus N under a launcher with a different WORLD_SIZE must not report a number (exit 2), whatever else
    is available on the box"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 2 and "refusing" in res.stderr


def test_dense_join_column_division_bound():
    """csrc/mash_distance.hip rowjoin_dense_kernel: field = column / ndw by one multiply-high with kmul = ceil(2^32 / ndw);
    exact for column < PER * ndw as long as PER * ndw^2 < 2^32 -- the host caps a stripe at 37,832 dwords (three 10-bit
    fields) / 46,328 dwords (two 16-bit fields) for that reason.  Checked at every multiple of 8 up to the caps, on the
    columns where a floor can go wrong (just below and at every multiple of ndw), and that the caps are tight."""
    def ok(ndw, per):
        kmul = ((1 << 32) + ndw - 1) // ndw
        cols = [m * ndw + d for m in range(1, per + 1) for d in (-1, 0) if m * ndw + d < per * ndw]
        return all((c * kmul) >> 32 == c // ndw for c in cols)
    assert all(ok(n, 3) for n in range(8, 37832 + 1, 8))
    assert all(ok(n, 2) for n in range(8, 46328 + 1, 8))
    assert not all(ok(n, 3) for n in range(37840, 40000, 8))  # beyond the cap the trick does fail somewhere


def test_level1_sketch_division_and_stage_swizzle():
    """csrc/mash_distance.hip coarse_scatter_staged_kernel: a batch's flat item index i (< 8192) is split into sketch
    i / s and element i % s by one multiply-high with sinv = ceil(2^32 / s), for EVERY s the staged scatter takes (1 < s <=
    8192; s == 1 is special-cased); the stage-slot swizzle pos ^ ((pos >> 5) & 7) is a permutation that keeps every
    aligned group of 8 slots in place (a run written out in order reads back its own slots)."""
    i = np.arange(8192, dtype=np.uint64)
    for s in range(2, 8193):
        sinv = np.uint64(((1 << 32) + s - 1) // s) & np.uint64(0xFFFFFFFF)
        assert ((i * sinv) >> np.uint64(32) == i // np.uint64(s)).all(), s
    pos = np.arange(8192, dtype=np.uint32)
    swz = pos ^ ((pos >> 5) & 7)
    assert sorted(swz.tolist()) == pos.tolist()
    assert ((swz >> 3) == (pos >> 3)).all()


def test_feeder_chunk_masks_by_one_multiply():
    """csrc/read_feeders.hip newline_mask16: the "byte is not X" flags of a dword (0x01 per byte, bits 0 / 8 / 16 / 24) are
    gathered into a nibble by ONE multiply with 0x00204081 (nibble at bits 21..24), a second set of flags shifted up by 4
    rides along (nibble at bits 25..28); ':' ';' '>' '?' are the bytes that equal 0x3A once bits 0 and 2 are cleared."""
    rng = np.random.default_rng(3)
    w = rng.integers(0, 1 << 32, 200_000, dtype=np.uint64)
    special = np.frombuffer(b"\n>;:?\x0b\x1a\x3c\x7e\xba", np.uint8)
    for k in range(4):      # plant the bytes of interest
        hit = rng.random(len(w)) < 0.3
        w[hit] = (w[hit] & ~np.uint64(0xFF << (8 * k))) | (rng.choice(special, int(hit.sum())).astype(np.uint64) << np.uint64(8 * k))

    def nonzero_bytes(x):
        return ((((x & np.uint64(0x7F7F7F7F)) + np.uint64(0x7F7F7F7F)) | x) >> np.uint64(7)) & np.uint64(0x01010101)

    z = nonzero_bytes(w ^ np.uint64(0x0A0A0A0A)) | (nonzero_bytes((w & np.uint64(0xFAFAFAFA)) ^ np.uint64(0x3A3A3A3A)) << np.uint64(4))
    g = (z * np.uint64(0x00204081)) & np.uint64(0xFFFFFFFF)
    not_nl, not_sp = (g >> np.uint64(21)) & np.uint64(0xF), (g >> np.uint64(25)) & np.uint64(0xF)
    for k in range(4):
        byte = (w >> np.uint64(8 * k)) & np.uint64(0xFF)
        assert ((((not_nl >> np.uint64(k)) & np.uint64(1)) == 0) == (byte == 0x0A)).all()
        assert ((((not_sp >> np.uint64(k)) & np.uint64(1)) == 0) == np.isin(byte, [0x3A, 0x3B, 0x3E, 0x3F])).all()


def test_fasta_two_scans_in_one_pass_combination_rule():
    """csrc/read_feeders.hip scan2_*: the per-line scans of (is_header, seq_len) run as ONE pass over segments of whole
    1024-line chunks although seq_len counts only behind the file's first header.  Every segment reports (headers, bytes
    behind its own first header, bytes in front of it, index of that header); the single-workgroup step in the middle
    finds the file's first header and takes from a segment: nothing (it lies in front), the bytes behind its own first
    header (it holds the file's first header), or everything (it starts behind it).  Restated in numpy against the plain
    definition: dst[k] = sum of seq_len[j] for first_header <= j < k."""
    SEGS = 512
    rng = np.random.default_rng(17)
    for n, first in [(0, None), (5, None), (5, 0), (5000, 4999), (5000, None), (600_000, 0), (600_000, 1023), (600_000, 1024),
                     (600_000, 1025), (600_000, 2047), (600_000, 300_000), (600_000, 599_999), (1_100_000, 777_777)]:
        is_header = np.zeros(n, np.uint64)
        if first is not None:
            is_header[first] = 1
            later = rng.random(n) < 0.02
            later[: first + 1] = False
            is_header[later] = 1
        seq_len = rng.integers(0, 200, n).astype(np.uint64)
        seq_len[is_header == 1] = 0
        per = ((n + SEGS - 1) // SEGS + 1023) // 1024 * 1024
        cnt, after, before, firstl = [], [], [], []
        for t in range(SEGS):
            lo, hi = min(n, t * per), min(n, t * per + per)
            h = np.flatnonzero(is_header[lo:hi])
            f = lo + int(h[0]) if len(h) else None
            cnt.append(int(is_header[lo:hi].sum()))
            firstl.append(f)
            cut = hi if f is None else f
            before.append(int(seq_len[lo:cut].sum()))
            after.append(int(seq_len[cut:hi].sum()))
        gfirst = min((f for f in firstl if f is not None), default=None)
        eff = []
        for t in range(SEGS):
            lo = min(n, t * per)
            if gfirst is None:
                eff.append(0)
            elif lo > gfirst:
                eff.append(after[t] + before[t])
            elif firstl[t] == gfirst:
                eff.append(after[t])
            else:
                eff.append(0)
        seg_start = np.concatenate([[0], np.cumsum(eff)])
        masked = seq_len.copy()
        masked[: (n if gfirst is None else gfirst)] = 0
        want = np.concatenate([[0], np.cumsum(masked)])
        assert gfirst == first
        for t in range(SEGS):
            lo = min(n, t * per)
            assert int(seg_start[t]) == int(want[lo]), (n, first, t)
        assert int(seg_start[SEGS]) == int(want[n])


def _plan4(maxval, nbk_log2, s, slice_len, cpp_max):
    """csrc/mash_distance.hip plan4_kernel, restated: None = the two-level build"""
    bits = max(maxval, 1).bit_length()
    shift = bits - nbk_log2 if bits > nbk_log2 else 0
    if bits < 17 or bits > 30 or shift < 4 or shift > 10:
        return None
    nce = (maxval >> 16) + 1
    R = min(max((s + slice_len - 1) // slice_len, 1), 64)
    cpp = min(max((nce + R - 1) // R, 2), cpp_max)
    R = (nce + cpp - 1) // cpp
    return dict(nc=1 << (bits - 16), R=R, cpp=cpp, magic=((1 << 32) + cpp - 1) // cpp, shift=shift, nce=nce)


def test_geometry_and_slice_bounds_of_the_sliced_index_build():
    """csrc/mash_distance.hip, round 5 (plan4_kernel, b4_part, check4_kernel's `pos`, fine4_kernel's lane permutation),
    restated in plain integers.  (a) the geometry: the parts cover every coarse bucket that can hold an item, at most 64
    parts (the pos table's row) of at most cpp_max coarse buckets (level 1's LDS counters), and the part of a coarse bucket by
    ONE multiply-high equals the division for every coarse bucket below 2^16.  (b) the kernel writes pos[r] from the
    TRANSITIONS a lane sees (element e in part ra, element e + 1 in part rb > ra: pos[ra + 1 .. rb] = e + 1; the first
    element: pos[0 .. its part] = 0; the last: pos[its part + 1 .. R] = s): equal to the definition pos[r] = number of
    elements below part r, and slice r holds exactly the elements of part r.  (c) the lane-to-item mapping of level 2 is a
    permutation of the workgroup's threads whose waves take eight pieces of 8 consecutive items T / 8 apart."""
    rng = np.random.default_rng(8)
    for it in range(400):
        bits = int(rng.integers(12, 33))
        maxval = int(rng.integers(1 << (bits - 1), 1 << bits)) if bits < 33 else 0xFFFFFFFF
        maxval = min(maxval, 0xFFFFFFFF)
        nbk_log2 = int(rng.integers(11, 25))
        s = int(rng.integers(1, 1025))
        slice_len = int(rng.choice([16, 20, 41, 83, 250, 1024]))
        cpp_max = int(rng.choice([512, 1024]))
        g = _plan4(maxval, nbk_log2, s, slice_len, cpp_max)
        if g is None:
            continue
        assert g["nc"] >= g["nce"] and g["nc"] <= 16384 and 16 - g["shift"] <= 12
        assert g["R"] * g["cpp"] >= g["nce"] and 1 <= g["R"] <= 64 and 2 <= g["cpp"] <= cpp_max
        coarse = np.arange(0, min(g["nc"], 65536), dtype=np.uint64)
        part = np.minimum((coarse * np.uint64(g["magic"])) >> np.uint64(32), g["R"] - 1)
        assert (part == np.minimum(coarse // np.uint64(g["cpp"]), g["R"] - 1)).all(), (it, g)
        # (b) a sketch below maxval, sometimes crowded into a few parts
        R = g["R"]
        if it % 3 == 0:
            hi = int(rng.integers(1, maxval + 1))
            x = np.sort(rng.integers(hi // 2, hi + 1, s, dtype=np.uint64))
        else:
            x = np.sort(rng.integers(0, maxval + 1, s, dtype=np.uint64))
        px = np.minimum(((x >> np.uint64(16)) * np.uint64(g["magic"])) >> np.uint64(32), R - 1).astype(np.int64)
        pos = np.full(R + 1, -1, np.int64)
        for e in range(s):  # every "lane" on its own, as the kernel
            ra = px[e]
            if e == 0:
                pos[0:ra + 1] = 0
            if e + 1 < s:
                rb = px[e + 1]
                if rb > ra:
                    pos[ra + 1:rb + 1] = e + 1
            else:
                pos[ra + 1:R + 1] = s
        want = np.array([int((px < r).sum()) for r in range(R + 1)])
        assert (pos == want).all(), (it, s, g)
        for r in range(R):
            sl = x[pos[r]:pos[r + 1]]
            assert (px[pos[r]:pos[r + 1]] == r).all()
            assert ((sl >> np.uint64(16)) >= r * g["cpp"]).all() and ((sl >> np.uint64(16)) < (r + 1) * g["cpp"]).all() or r == R - 1
    # config 3's own numbers (100,000 sketches of 1000, 2^23 buckets): what the profiles quote
    g = _plan4((7370 << 16) - 1, 23, 1000, 41, 1024)
    assert (g["nc"], g["R"], g["cpp"], g["shift"]) == (8192, 25, 295, 6)  # polyhip_mash_index_build_info_dev on the GPU box
    # (c) fine4_kernel's ptid for T = 1024 and 512
    for T in (1024, 512):
        tid = np.arange(T)
        ptid = ((((tid >> 3) & 7) * (T // 64) + (tid >> 6)) << 3) | (tid & 7)
        assert sorted(ptid.tolist()) == list(range(T))
        for w in range(T // 64):
            pieces = sorted(set((ptid[w * 64:(w + 1) * 64] >> 3).tolist()))
            assert len(pieces) == 8 and all(b - a == T // 64 for a, b in zip(pieces, pieces[1:]))


def test_sliced_index_build_restated_end_to_end():
    """The sliced index build of csrc/mash_distance.hip (round 5) as an executable specification in numpy: the same geometry
    (_plan4), the same intermediate item [hash & 0xFFFF : 16 | id & 0xFFFF : 16] in the same places (coarse bucket = hash >> 16,
    segment = id >> 16, level 1 = any order inside a segment), the same compact item
    [hash's bits below its fine bucket : shift | occurrence number + 1 : 11 - shift | counter dword : 16 | field shift : 5],
    the repeated hashes numbered behind the placement from the logged records.  Decoding the result must give back exactly
    the multiset {(hash, sketch, occurrence number)} of the input -- what the join relies on (same value AND occurrence number
    < multiplicity, the counter of column `sketch`)."""
    rng = np.random.default_rng(55)
    CK_LOW = 21
    for ny, s, bits, nbk_log2, field_bits in ((700, 64, 26, 17, 10), (70_000 // 50, 96, 24, 16, 10), (1300, 128, 23, 15, 16)):
        # sketches: ascending, some hashes repeated inside a sketch, ids on both sides of 65,536 for one configuration
        id0 = 65_000 if ny == 1400 else 0                       # (the ids the items carry: a window of a larger set)
        Y = np.sort(rng.integers(0, 1 << bits, (ny, s), dtype=np.uint64), axis=1)
        for q in range(0, ny, 9):
            e = int(rng.integers(1, s))
            Y[q, e] = Y[q, e - 1]
        for q in range(3, ny, 40):
            Y[q, 5:5 + 1 + q % 3] = Y[q, 5]
        Y.sort(axis=1)
        maxval = int(Y[:, -1].max())
        g = _plan4(maxval, nbk_log2, s, 41, 1024)
        assert g is not None
        shift, nc = g["shift"], g["nc"]
        fpc_log2 = 16 - shift
        per = 32 // field_bits
        ncols = id0 + ny
        ndw = (((ncols + per - 1) // per) + 7) & ~7
        G = 2 if ncols > 65536 else 1
        # ---- check pass: histogram per (coarse, group), the list of repeated hashes
        ids = id0 + np.arange(ny)
        hist = np.zeros((nc, G), np.int64)
        dups = []
        for q in range(ny):
            np.add.at(hist[:, ids[q] >> 16], (Y[q] >> np.uint64(16)).astype(np.int64), 1)
            e = 0
            while e < s:
                a = 1
                while e + a < s and Y[q, e + a] == Y[q, e]:
                    a += 1
                for k in range(1, a):
                    dups.append((int(Y[q, e]), int(ids[q]), k))
                e += a
        c4start = np.concatenate([[0], np.cumsum(hist.reshape(-1))])
        # ---- level 1: every item to its (coarse, group) segment, in a shuffled order inside it
        cur = c4start[:-1].copy()
        inter = np.zeros(ny * s, np.uint32)
        order = rng.permutation(ny * s)
        flatv, flati = Y.reshape(-1), np.repeat(ids, s)
        for t in order:
            v, i = int(flatv[t]), int(flati[t])
            seg = (v >> 16) * G + (i >> 16)
            inter[cur[seg]] = ((v & 0xFFFF) << 16) | (i & 0xFFFF)
            cur[seg] += 1
        assert (cur == c4start[1:]).all()
        # ---- level 2: per coarse bucket, counting sort by fine bucket; compact items with occurrence number 0
        kmul = ((1 << 32) + ndw - 1) // ndw
        items = np.zeros(ny * s, np.uint32)
        start = np.zeros((nc << fpc_log2) + 1, np.int64)
        low_mask = (1 << shift) - 1

        def compact_of(rem16, col, occ1):
            k = (col * kmul) >> 32
            assert k == col // ndw
            return ((rem16 & low_mask) << (32 - shift)) | (occ1 << CK_LOW) | ((col - k * ndw) << 5) | (k * field_bits)
        for c in range(nc):
            lo, mid, hi = c4start[c * G], c4start[c * G + G - 1], c4start[c * G + G]
            it = inter[lo:hi].astype(np.int64)
            fine = it >> (16 + shift)
            cnt = np.bincount(fine, minlength=1 << fpc_log2)
            st = lo + np.concatenate([[0], np.cumsum(cnt)])[:-1]
            start[(c << fpc_log2):((c + 1) << fpc_log2)] = st
            cursor = st.copy()
            for p in range(hi - lo):
                col = int(it[p] & 0xFFFF) | (65536 if (G == 2 and lo + p >= mid) else 0)
                items[cursor[fine[p]]] = compact_of(int(it[p] >> 16), col, 1)
                cursor[fine[p]] += 1
        start[nc << fpc_log2] = ny * s
        occ_mask = ((1 << (11 - shift)) - 1) << CK_LOW
        for v, i, k in dups:  # the number-th (0-based) of the equal items of the fine bucket, in slot order, gets the number
            fb = v >> shift
            want = compact_of(v & 0xFFFF, i, 1)
            seen = 0
            for at in range(start[fb], start[fb + 1]):
                if ((int(items[at]) ^ want) & ~occ_mask & 0xFFFFFFFF) == 0:
                    if seen == k:
                        items[at] = (want & ~occ_mask) | ((k + 1) << CK_LOW)
                        break
                    seen += 1
            else:
                raise AssertionError("a logged copy without a slot")
        # ---- decode: the multiset of (hash, sketch, occurrence number)
        got = []
        for fb in range(nc << fpc_log2):
            for at in range(start[fb], start[fb + 1]):
                w = int(items[at])
                low = w >> (32 - shift)
                occ = ((w & occ_mask) >> CK_LOW) - 1
                dword, fsh = (w >> 5) & 0xFFFF, w & 31
                col = (fsh // field_bits) * ndw + dword
                got.append(((fb << shift) | low, col, occ))
        want = []
        for q in range(ny):
            e = 0
            while e < s:
                a = 1
                while e + a < s and Y[q, e + a] == Y[q, e]:
                    a += 1
                want += [(int(Y[q, e]), int(ids[q]), k) for k in range(a)]
                e += a
        assert sorted(got) == sorted(want), (ny, s, bits)
