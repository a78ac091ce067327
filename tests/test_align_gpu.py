"""K3 parity: poly_amd.align (HIP, through the C ABI) vs the CPU oracle.
Score, argmax position and the alphabet-error symbol must be identical.

Mirrors search/align/align_test.go:139-292 and example_test.go:49-111 where the
reference has a test, then widens to batches on both kernel families."""
import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def al():
    from poly_amd import align, alphabet, matrix
    return align, alphabet, matrix


def _scoring(al, symbols, scores, gap):
    align, alphabet, matrix = al
    a = alphabet.NewAlphabet(list(symbols))
    return align.NewScoring(matrix.NewSubstitutionMatrix(a, a, scores), gap)


def _pack(seqs):
    offs = np.zeros(len(seqs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    return np.frombuffer(b"".join(seqs), np.uint8).copy(), offs


def _oracle_batch(reads, refs, omat, gap):
    out = []
    for a, b in zip(reads, refs):
        try:
            sc, _, _, ea, eb = orc.smith_waterman(a, b, omat, gap)
            out.append((sc, ea, eb, 0))
        except orc.AlphabetError as e:
            sym = str(e).split(" ")[1]
            which = 1 if (len(a) and (not _in(omat.first, a[0:1]) or (all(_in(omat.second, bytes([c])) for c in b)))) else 2
            out.append((0, 0, 0, (which << 8) | ord(sym)))
    return out


def _in(alpha, ch):
    return len(ch) == 1 and ch[0] < 0x80 and chr(ch[0]) in alpha


def _check(al, scoring, omat, gap, reads, ref=None, refs=None, expect_path=None):
    """expect_path 1 = "a shared-reference fast path": the batch is run through every variant that
    qualifies -- one wave per pair (4, small batches), packed (3), lane per pair (1) -- by switching the
    others off with POLYHIP_SW_WAVE / POLYHIP_SW_PACKED, and each must equal the oracle.  expect_path 2 =
    everything else: the per-pair-B register-tiled kernel (5) or the one-wave-per-pair kernel (6: long reads, gap >= 0,
    wide scores) where they qualify, and the generic kernel."""
    import os
    align = al[0]
    A, offA = _pack(reads)
    if refs is None:
        B, offB = _pack([ref])[0], None
        want = _oracle_batch(reads, [ref] * len(reads), omat, gap)
    else:
        B, offB = _pack(refs)
        want = _oracle_batch(reads, refs, omat, gap)
    variants = [({}, None)]
    if expect_path == 1:
        variants = [({}, (1, 3, 4)), ({"POLYHIP_SW_WAVE": "0"}, (1, 3)),
                    ({"POLYHIP_SW_WAVE": "0", "POLYHIP_SW_PACKED": "0"}, (1,))]
    elif expect_path == 2:  # "not a shared-reference fast path": per-pair register-tiled kernel (5) or generic (2)
        variants = [({}, (2, 5, 6)), ({"POLYHIP_SW_PAIR": "0"}, (2, 6)), ({"POLYHIP_SW_WAVE": "0"}, (2, 5)),
                    ({"POLYHIP_SW_PAIR": "0", "POLYHIP_SW_WAVE": "0"}, (2,))]
    for env, paths in variants:
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            got = align.sw_batch_packed(scoring, A, offA, B, offB)
            path = align.last_path()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        if paths is not None:
            assert path in paths, (path, paths)
        elif expect_path is not None:
            assert path == expect_path
        for p, w in enumerate(want):
            g = (int(got[0][p]), int(got[1][p]), int(got[2][p]), int(got[3][p]))
            assert g == w, f"path {path} pair {p}: got {g} want {w} (lenA {len(reads[p])})"


def _mutate(rng, seq: bytes, sub=0.05, indel=0.01) -> bytes:
    out = bytearray()
    for c in seq:
        r = rng.random()
        if r < indel / 2:
            continue
        if r < indel:
            out.append(int(rng.choice(list(b"ACGT"))))
        out.append(int(rng.choice(list(b"ACGT"))) if rng.random() < sub else c)
    return bytes(out)


MAT3 = [[0, 0, 0, 0, 0], [0, 3, -3, -3, -3], [0, -3, 3, -3, -3], [0, -3, -3, 3, -3], [0, -3, -3, -3, 3]]


def test_TestSmithWaterman_scores(al):
    """search/align/align_test.go:139-292 (scores + the argmax the traceback starts from)"""
    sc = _scoring(al, "-ACGT", MAT3, -2)
    om = orc.SubstitutionMatrix("-ACGT", "-ACGT", MAT3)
    cases = [(b"TGTTACGG", b"GGTTGACTA", 13), (b"ACACACTA", b"AGCACACA", 17), (b"", b"GAT", 0), (b"", b"", 0),
             (b"G", b"A", 0), (b"G", b"G", 3), (b"G", b"GATTACA", 3)]
    for a, b, want in cases:
        got = al[0].sw_batch_packed(sc, *_pack([a]), _pack([b])[0], None)
        assert int(got[0][0]) == want, (a, b)
        _check(al, sc, om, -2, [a], ref=b)
        _check(al, sc, om, -2, [a], refs=[b])  # generic kernel, same answer


def test_examples(al):
    """search/align/example_test.go:49-111"""
    pm1 = (2 * np.eye(5, dtype=int) - 1).tolist()
    sc = _scoring(al, "ACGTU", pm1, -1)
    got = al[0].sw_batch_packed(sc, *_pack([b"GATTACA"]), _pack([b"GCATGCU"])[0], None)
    assert int(got[0][0]) == 2
    # NUC_4 indexed by a mis-ordered alphabet {A,C,G,T,-}: 'A' hits the all-zero row
    sc = _scoring(al, "ACGT-", al[2].NUC_4, -1)
    got = al[0].sw_batch_packed(sc, *_pack([b"GATTACA"]), _pack([b"GCATGCT"])[0], None)
    assert int(got[0][0]) == 15
    _check(al, sc, orc.SubstitutionMatrix("ACGT-", "ACGT-", orc.NUC_4_SCORES), -1, [b"GATTACA"], ref=b"GCATGCT")


def test_config4_shape_fast_kernel(al):
    """BASELINE config 4 shape: 150 bp reads (5 % subs, 1 % indels) vs one 5 kb reference, NUC_4, gap -2"""
    rng = np.random.default_rng(0xC4)
    ref = orc.synth_dna(0xC4, 5000).tobytes()
    reads = []
    for _ in range(300):
        p = int(rng.integers(0, 5000 - 150))
        reads.append(_mutate(rng, ref[p:p + 150])[:152])
    reads += [orc.synth_dna(99, 150).tobytes(), b"", b"A", ref[:152], ref[-150:]]
    sc = _scoring(al, "-ACGT", al[2].NUC_4, -2)
    om = orc.SubstitutionMatrix("-ACGT", "-ACGT", orc.NUC_4_SCORES)
    _check(al, sc, om, -2, reads, ref=ref, expect_path=1)


@pytest.mark.parametrize("maxlen,reflen", [(64, 300), (152, 1024), (152, 1027), (256, 2100), (40, 3), (152, 4), (10, 1)])
def test_ragged_fast_kernel(al, maxlen, reflen):
    rng = np.random.default_rng(maxlen * 7 + reflen)
    ref = orc.synth_dna(1234 + reflen, reflen).tobytes()
    reads = [orc.synth_dna(int(rng.integers(1, 1 << 30)), int(rng.integers(0, maxlen + 1))).tobytes() for _ in range(300)]
    reads[0] = orc.synth_dna(5, maxlen).tobytes()
    # plant near-copies so that long, high-scoring alignments and ties exist
    for i in range(1, 60):
        L = int(rng.integers(1, min(maxlen, reflen) + 1))
        p = int(rng.integers(0, reflen - L + 1))
        reads[i] = _mutate(rng, ref[p:p + L], 0.03, 0.02)[:maxlen]
    pm = [[0, 0, 0, 0, 0], [0, 2, -1, -1, -1], [0, -1, 2, -1, -1], [0, -1, -1, 2, -1], [0, -1, -1, -1, 2]]
    sc = _scoring(al, "-ACGT", pm, -1)
    om = orc.SubstitutionMatrix("-ACGT", "-ACGT", pm)
    _check(al, sc, om, -1, reads, ref=ref, expect_path=1)


def test_tie_breaking_repeats(al):
    """many co-optimal cells: the first maximum in row-major order must win (align.go:197)"""
    ref = (b"ACGT" * 300)[:1100]
    reads = [b"ACGT" * k for k in range(1, 30)] + [b"CGTA" * 5, b"TTTT", b"GTAC" * 30, b"A"]
    sc = _scoring(al, "-ACGT", MAT3, -2)
    om = orc.SubstitutionMatrix("-ACGT", "-ACGT", MAT3)
    _check(al, sc, om, -2, reads, ref=ref, expect_path=1)
    _check(al, sc, om, -2, reads, refs=[ref] * len(reads), expect_path=2)


def test_default_matrix_protein_like(al):
    """matrix.Default (26 letters, +1/-1), gap -1 -> CP=32 instantiation"""
    rng = np.random.default_rng(26)
    letters = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ", np.uint8)
    ref = rng.choice(letters, 900).tobytes()
    reads = [rng.choice(letters, int(rng.integers(0, 150))).tobytes() for _ in range(200)]
    for i in range(40):
        p = int(rng.integers(0, 800))
        reads[i] = ref[p:p + int(rng.integers(5, 100))]
    sc = al[0].NewScoring(None, -1)
    _check(al, sc, orc.DEFAULT_MATRIX, -1, reads, ref=ref, expect_path=1)


def test_generic_kernel_long_and_per_pair(al):
    rng = np.random.default_rng(77)
    sc = _scoring(al, "-ACGT", MAT3, -2)
    om = orc.SubstitutionMatrix("-ACGT", "-ACGT", MAT3)
    ref = orc.synth_dna(31, 700).tobytes()
    # A longer than the register tile -> generic
    reads = [orc.synth_dna(100 + i, int(rng.integers(257, 500))).tobytes() for i in range(20)]
    reads[0] = _mutate(rng, ref[100:450])
    _check(al, sc, om, -2, reads, ref=ref, expect_path=2)
    # per-pair B of ragged lengths
    refs = [orc.synth_dna(500 + i, int(rng.integers(0, 300))).tobytes() for i in range(64)]
    reads = [orc.synth_dna(900 + i, int(rng.integers(0, 200))).tobytes() for i in range(64)]
    _check(al, sc, om, -2, reads, refs=refs, expect_path=2)
    # odd scoring: positive gap, large scores, asymmetric matrix
    asym = [[0, 0, 0, 0, 0], [0, 300, -7, 2, -1], [0, -100, 250, 0, 3], [0, 5, -3, 400, -9], [0, 1, 2, -300, 200]]
    sc2 = _scoring(al, "-ACGT", asym, 1)
    om2 = orc.SubstitutionMatrix("-ACGT", "-ACGT", asym)
    _check(al, sc2, om2, 1, reads[:16], ref=ref[:120], expect_path=2)


def test_asymmetric_two_alphabets_fast(al):
    align, alphabet, matrix = al
    rows, cols = "ACGT", "ACGTN"
    scores = [[4, -2, -1, -3, 0], [-2, 5, -3, -1, 0], [-1, -4, 6, -2, 0], [-3, -1, -2, 3, 0]]
    sc = align.NewScoring(matrix.NewSubstitutionMatrix(alphabet.NewAlphabet(list(rows)), alphabet.NewAlphabet(list(cols)), scores), -3)
    om = orc.SubstitutionMatrix(rows, cols, scores)
    rng = np.random.default_rng(3)
    ref = bytes(rng.choice(list(b"ACGTN"), 500).tolist())
    reads = [bytes(rng.choice(list(b"ACGT"), int(rng.integers(0, 64))).tolist()) for _ in range(100)]
    _check(al, sc, om, -3, reads, ref=ref, expect_path=1)


def test_error_symbol_order(al):
    """align.go:189-191 + matrix.go:29-36"""
    sc = _scoring(al, "-ACGT", MAT3, -2)
    om = orc.SubstitutionMatrix("-ACGT", "-ACGT", MAT3)
    reads = [b"XG", b"GX", b"GGGGXGY", b"", b"ACGT", b"NACGT", b"ACGTacgt"]
    for ref in (b"GY", b"GA", b"", b"ZZZ", b"ACGTNNACGT"):
        _check(al, sc, om, -2, reads, ref=ref, expect_path=1)
        _check(al, sc, om, -2, reads, refs=[ref] * len(reads), expect_path=2)
    # bytes >= 0x80 are never in a one-byte-symbol alphabet (string(byte) is 2 bytes of UTF-8)
    got = al[0].sw_batch_packed(sc, *_pack([b"AC\xc3G"]), _pack([b"ACG"])[0], None)
    assert int(got[3][0]) == (1 << 8) | 0xC3 and int(got[0][0]) == 0


@pytest.mark.parametrize("n", [4096, 1_000_000])
def test_device_resident_config4_sample(al, n):
    """device-resident entry point + size-independent property at full read size (and, n = 1,000,000, at
    BASELINE configs[3]'s full batch size): an exact substring of the reference scores 5*len and ends where
    it was cut"""
    import torch
    align = al[0]
    dev = torch.device("cuda:0")
    ref = orc.synth_dna(0xC4, 5000)
    L = 150
    rng = np.random.default_rng(1)
    starts = rng.integers(0, 5000 - L, n)
    reads = np.stack([ref[s:s + L] for s in starts])
    sc = _scoring(al, "-ACGT", al[2].NUC_4, -2)
    A = torch.from_numpy(reads.reshape(-1).copy()).to(dev)
    offA = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    B = torch.from_numpy(ref.copy()).to(dev)
    score = torch.zeros(n, dtype=torch.int64, device=dev)
    ea = torch.zeros(n, dtype=torch.int32, device=dev)
    eb = torch.zeros(n, dtype=torch.int32, device=dev)
    er = torch.zeros(n, dtype=torch.int32, device=dev)
    work = torch.empty(align.sw_workspace_bytes(sc, n, L, 5000), dtype=torch.uint8, device=dev)
    align.sw_batch_dev(sc, A, offA, L, B, None, 5000, score, ea, eb, er, work)
    torch.cuda.synchronize()
    assert align.last_path() in (1, 3, 4)
    assert (score.cpu().numpy() == 5 * L).all()
    assert (ea.cpu().numpy() == L).all()
    assert (er.cpu().numpy() == 0).all()
    # first row-major maximum: the earliest occurrence of the read in the reference
    refb = ref.tobytes()
    sample = range(n) if n <= 4096 else range(0, n, 53)
    first = np.array([refb.find(reads[i].tobytes()) + L for i in sample])
    assert (eb.cpu().numpy()[list(sample)] == first).all()


@pytest.mark.parametrize("kind", ["random", "repeats", "short_ref", "bad_symbols", "long_ref", "random_250", "repeats_250",
                                  "bad_symbols_250", "random_100", "repeats_100"])
def test_packed_pass_equals_exact_kernel(al, monkeypatch, kind):
    """The packed two-pairs-per-lane pass + locate + tie list (path 3) against the exact 32-bit kernel alone
    (POLYHIP_SW_PACKED=0, path 1) on 120k ragged reads at 0..90 % substitutions: score, endA, endB and err
    of every pair are equal.  `repeats`: the reference is a 16-fold tandem repeat with a few point changes,
    so most maxima occur in several blocks (ties -> the exact kernel decides which is first in row-major
    order); `short_ref`: fewer columns than one LDS chunk; `bad_symbols`: reads and reference with bytes
    outside the alphabets; `long_ref`: a 15 kb reference (the locate kernel's byte profile takes 120 KB of LDS).
    `_250`: up to 256 rows (250-bp reads), the one-workgroup-per-CU instantiation whose H rows spill into AGPRs.
    A sample is also checked against the oracle."""
    import torch
    align = al[0]
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    long_reads = kind.endswith("_250")
    short_reads = kind.endswith("_100")  # the longest read leaves row groups unused: sw_pk_kernel<152, true> skips them per wave
    kind = kind.replace("_250", "").replace("_100", "")
    LB = {"random": 5000, "repeats": 4800, "short_ref": 37, "bad_symbols": 2000, "long_ref": 15000}[kind]  # long_ref: 120 KB of LDS
    ref = orc.synth_dna(0xC4, LB).copy()
    if kind == "repeats":
        unit = ref[:300].copy()
        ref = np.tile(unit, 16)
        ref[rng.integers(0, LB, 12)] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 12)]
    n, L = {"long_ref": 50_000, "short_ref": 120_001, "bad_symbols": 99_999}.get(kind, 120_000), 152  # odd counts: a lane's second pair may be missing
    if long_reads:
        n, L = n // 2 + 1, 256
    if short_reads:
        L = 100
    starts = rng.integers(0, max(1, LB - L), n)
    idx = (starts[:, None] + np.arange(L)[None, :]) % LB
    reads = ref[idx]
    rate = np.linspace(0.0, 0.9, n)[:, None]
    hit = rng.random((n, L)) < rate
    reads[hit] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(hit.sum()))]
    lens = rng.integers(0, L + 1, n)
    lens[rng.random(n) < 0.5] = 250 if long_reads else 98 if short_reads else 150
    if kind == "bad_symbols":
        bad = rng.random(n) < 0.02
        reads[bad, rng.integers(0, 20, int(bad.sum()))] = ord("N")
    offs = np.zeros(n + 1, np.int64)
    offs[1:] = np.cumsum(lens)
    flat = np.concatenate([reads[i, :lens[i]] for i in range(n)]) if n else np.zeros(0, np.uint8)
    sc = _scoring(al, "-ACGT", al[2].NUC_4, -2)
    A = torch.from_numpy(flat.copy()).to(dev)
    offA = torch.from_numpy(offs).to(dev)

    lanes_seen = []

    def run(refarr, packed, half=True, fixed_slot=True, two_lanes=True):
        if two_lanes:   # sw_pk1x2_kernel (round 6): a lane's 152 rows over two lanes, four waves per SIMD (the default at 152 rows)
            monkeypatch.delenv("POLYHIP_SW_PK1X2", raising=False)
        else:           # sw_pk1_kernel: one lane per two pairs, two waves per SIMD
            monkeypatch.setenv("POLYHIP_SW_PK1X2", "0")
        if packed:
            monkeypatch.delenv("POLYHIP_SW_PACKED", raising=False)
        else:
            monkeypatch.setenv("POLYHIP_SW_PACKED", "0")
        if half:
            monkeypatch.delenv("POLYHIP_SW_F16", raising=False)
        else:
            monkeypatch.setenv("POLYHIP_SW_F16", "0")
        if fixed_slot:  # sw_pk1_kernel: one wave per workgroup, the block's table at a fixed LDS address (the default)
            monkeypatch.delenv("POLYHIP_SW_PK1", raising=False)
        else:           # sw_pk_kernel: four waves share 36 KB chunks of the profile
            monkeypatch.setenv("POLYHIP_SW_PK1", "0")
        B = torch.from_numpy(refarr.copy()).to(dev)
        score = torch.full((n,), -7, dtype=torch.int64, device=dev)
        ea, eb, er = (torch.full((n,), -7, dtype=torch.int32, device=dev) for _ in range(3))
        work = torch.empty(align.sw_workspace_bytes(sc, n, L, len(refarr)), dtype=torch.uint8, device=dev)
        align.sw_batch_dev(sc, A, offA, L, B, None, len(refarr), score, ea, eb, er, work)
        torch.cuda.synchronize()
        monkeypatch.delenv("POLYHIP_SW_PK1", raising=False)
        monkeypatch.delenv("POLYHIP_SW_PK1X2", raising=False)
        lanes_seen.append(align.last_packed_lanes())
        return [t.cpu().numpy() for t in (score, ea, eb, er)], align.last_path(), align.last_packed_half()

    refs = [ref]
    if kind == "bad_symbols":
        r2 = ref.copy()
        r2[1234] = ord("N")
        refs.append(r2)
    om = orc.SubstitutionMatrix("-ACGT", "-ACGT", orc.NUC_4_SCORES)
    for refarr in refs:
        # up to 152 rows at NUC.4's 5 per match stay below 2048: the half-float cell of gfx950 is the default there,
        # the int16 cell (POLYHIP_SW_F16=0) and the exact 32-bit kernel are both run beside it
        got, path, half = run(refarr, True)
        got16, path16, half16 = run(refarr, True, half=False)
        gotc, pathc, halfc = run(refarr, True, fixed_slot=False)
        got1, path1, half1 = run(refarr, True, two_lanes=False)
        want, path0, _ = run(refarr, False)
        assert (path, path16, pathc, path1, path0) == (3, 3, 3, 3, 1)
        assert (half, half16, halfc, half1) == (True, False, True, True)
        # which form of the packed pass ran: the default spreads a lane's rows over two lanes (65..152 rows) or four (256 rows
        # on the 64-row tile); the int16 cell and the chunk-staged kernel keep one lane, as does POLYHIP_SW_PK1X2=0; the exact
        # kernel is no packed pass
        assert lanes_seen[-5:] == ([4, 4, 4, 4, 0] if long_reads else [2, 1, 1, 1, 0]), lanes_seen
        for g, g16, gc, g1, w in zip(got, got16, gotc, got1, want):
            assert (g == w).all() and (g16 == w).all() and (gc == w).all() and (g1 == w).all()
        refb = refarr.tobytes()
        for p in range(0, n, 1501):
            a = flat[offs[p]:offs[p + 1]].tobytes()
            try:
                s, _, _, ea_, eb_ = orc.smith_waterman(a, refb, om, -2)
                assert (int(got[0][p]), int(got[1][p]), int(got[2][p]), int(got[3][p])) == (s, ea_, eb_, 0), p
            except orc.AlphabetError:
                assert int(got[3][p]) != 0 and int(got[0][p]) == 0


@pytest.mark.parametrize("smax,gap,expect_half", [(13, -1, True), (13, -2035, True), (14, -1, False), (13, -2036, False)])
def test_half_float_cell_limits(al, monkeypatch, smax, gap, expect_half):
    """The half-float packed cell holds integers below 2048 exactly: a matrix whose best score is 13 keeps 152 rows
    inside (13 * 152 = 1976 -- reads equal to a stretch of the reference reach exactly that), 14 does not and takes
    the int16 cell; smax + |gap| up to 2048 (the profile's first column of a block carries score + |gap|).  Every pair against the exact 32-bit kernel, a sample against the oracle."""
    import torch
    align = al[0]
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(smax * 7 + (-gap))
    n, L, LB = 60_001, 152, 1500
    scores = np.full((5, 5), -3, np.int64)
    np.fill_diagonal(scores, [smax, smax, smax - 1, smax - 2, 1])
    scores[1, 2] = scores[2, 1] = 2
    scores = scores.tolist()
    ref = orc.synth_dna(0x51, LB).copy()
    starts = rng.integers(0, LB, n)
    reads = ref[(starts[:, None] + np.arange(L)[None, :]) % LB]
    reads[0] = ord("A")  # with a poly-A stretch in the reference: the largest score a read of 152 can reach
    ref[700:700 + L] = ord("A")
    rate = np.repeat(np.linspace(0.0, 0.6, n)[:, None], L, 1)
    rate[:64] = 0.0
    hit = rng.random((n, L)) < rate
    reads[hit] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(hit.sum()))]
    lens = np.where(rng.random(n) < 0.5, L, rng.integers(0, L + 1, n))
    lens[:64] = L
    offs = np.zeros(n + 1, np.int64)
    offs[1:] = np.cumsum(lens)
    flat = np.concatenate([reads[i, :lens[i]] for i in range(n)])
    sc = _scoring(al, "-ACGT", scores, gap)
    om = orc.SubstitutionMatrix("-ACGT", "-ACGT", scores)
    A = torch.from_numpy(flat.copy()).to(dev)
    offA = torch.from_numpy(offs).to(dev)
    B = torch.from_numpy(ref.copy()).to(dev)

    def run(packed):
        if packed:
            monkeypatch.delenv("POLYHIP_SW_PACKED", raising=False)
        else:
            monkeypatch.setenv("POLYHIP_SW_PACKED", "0")
        score = torch.full((n,), -7, dtype=torch.int64, device=dev)
        ea, eb, er = (torch.full((n,), -7, dtype=torch.int32, device=dev) for _ in range(3))
        work = torch.empty(align.sw_workspace_bytes(sc, n, L, LB), dtype=torch.uint8, device=dev)
        align.sw_batch_dev(sc, A, offA, L, B, None, LB, score, ea, eb, er, work)
        torch.cuda.synchronize()
        return [t.cpu().numpy() for t in (score, ea, eb, er)], align.last_path(), align.last_packed_half()

    got, path, half = run(True)
    want, path0, _ = run(False)
    assert (path, path0, half) == (3, 1, expect_half)
    for g, w in zip(got, want):
        assert (g == w).all()
    assert int(got[0][0]) == smax * L
    refb = ref.tobytes()
    for p in list(range(0, 64, 9)) + list(range(64, n, 2503)):
        a = flat[offs[p]:offs[p + 1]].tobytes()
        s, _, _, ea_, eb_ = orc.smith_waterman(a, refb, om, gap)
        assert (int(got[0][p]), int(got[1][p]), int(got[2][p])) == (s, ea_, eb_), p


@pytest.mark.parametrize("maxA,LB,n,repeats", [(300, 3000, 30_001, False), (400, 2000, 20_001, True), (500, 2000, 17_000, True), (600, 1500, 14_001, False),
                                               (1000, 1500, 8_300, False), (1216, 1200, 7_001, True), (2048, 800, 4_200, False)])
def test_long_reads_packed_banded_pass_equals_wave_kernel(al, monkeypatch, maxA, LB, n, repeats):
    """Reads of 257..2048 rows against one reference, enough of them to fill the chip (path 7): the packed banded
    pass (K = 2..16 lanes per pair, every (rows per lane, K) tile once) + the wave kernel in locate mode against the
    wave kernel sweeping the whole reference (POLYHIP_SW_PACKED=0, path 6): score, endA, endB, err of every pair
    equal; a sample against the oracle.  Ragged lengths, 0..90 % substitutions, indels; `repeats`: a tandem-repeat
    reference, so maxima occur in several blocks (tie bit -> full sweep)."""
    import torch
    align = al[0]
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(maxA)
    ref = orc.synth_dna(0xC4, LB).copy()
    if repeats:
        unit = ref[:LB // 8].copy()
        ref = np.tile(unit, 8)[:LB]
        ref[rng.integers(0, LB, 6)] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 6)]
    starts = rng.integers(0, LB, n)
    starts[0] = 0  # read 0: an exact substring of full length
    idx = (starts[:, None] + np.arange(maxA)[None, :]) % LB
    reads = ref[idx]
    rate = np.linspace(0.0, 0.9, n)[:, None]
    hit = rng.random((n, maxA)) < rate
    reads[hit] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(hit.sum()))]
    # indels: drop one base at a random position in a third of the reads
    cut = rng.integers(1, maxA, n)
    drop = rng.random(n) < 0.33
    drop[0] = False
    col = np.arange(maxA)[None, :]
    src = np.where(drop[:, None] & (col >= cut[:, None]), np.minimum(col + 1, maxA - 1), col)
    reads = np.take_along_axis(reads, src, axis=1)
    lens = rng.integers(maxA // 3, maxA + 1, n)
    lens[0] = maxA
    lens[rng.random(n) < 0.01] = 0
    for p in range(5, n, 1009):  # a byte outside the alphabet in a few reads (first, last or any position): the error code
        if lens[p]:
            reads[p, [0, lens[p] - 1, int(rng.integers(0, lens[p]))][(p // 1009) % 3]] = ord("N")
    offs = np.zeros(n + 1, np.int64)
    offs[1:] = np.cumsum(lens)
    flat = np.concatenate([reads[i, :lens[i]] for i in range(n)])
    sc = _scoring(al, "-ACGT", al[2].NUC_4, -2)
    A = torch.from_numpy(flat.copy()).to(dev)
    offA = torch.from_numpy(offs).to(dev)
    B = torch.from_numpy(ref.copy()).to(dev)

    def run(packed, half=True):
        if packed:
            monkeypatch.delenv("POLYHIP_SW_PACKED", raising=False)
        else:
            monkeypatch.setenv("POLYHIP_SW_PACKED", "0")
        if half:
            monkeypatch.delenv("POLYHIP_SW_F16", raising=False)
        else:
            monkeypatch.setenv("POLYHIP_SW_F16", "0")
        score = torch.full((n,), -7, dtype=torch.int64, device=dev)
        ea, eb, er = (torch.full((n,), -7, dtype=torch.int32, device=dev) for _ in range(3))
        work = torch.empty(align.sw_workspace_bytes(sc, n, maxA, LB), dtype=torch.uint8, device=dev)
        align.sw_batch_dev(sc, A, offA, maxA, B, None, LB, score, ea, eb, er, work)
        torch.cuda.synchronize()
        return [t.cpu().numpy() for t in (score, ea, eb, er)], align.last_path(), align.last_packed_half()

    got, path, half = run(True)
    want, path0, _ = run(False)
    assert (path, path0) == (7, 6)
    assert half == (5 * min(maxA, LB) <= 2047)  # 300 and 400 rows stay below 2048: the half-float cell
    for g, w in zip(got, want):
        assert (g == w).all()
    assert int((want[3] != 0).sum()) >= n // 1009 - 1
    # up to 1024 rows the locate step runs on a byte profile of the pair (sw_wave8_kernel); POLYHIP_SW_WAVE8=0: the general
    # one-wave-per-pair kernel in locate mode
    monkeypatch.setenv("POLYHIP_SW_WAVE8", "0")
    got_t, path_t, _ = run(True)
    monkeypatch.delenv("POLYHIP_SW_WAVE8", raising=False)
    assert path_t == 7
    for g, w in zip(got_t, want):
        assert (g == w).all()
    if half:  # ... and the int16 cell on the same batch
        got16, path16, half16 = run(True, half=False)
        assert (path16, half16) == (7, False)
        for g, w in zip(got16, want):
            assert (g == w).all()
    # the default takes 64 rows per lane where a 64-row tile holds the read (round 6: four waves per SIMD);
    # POLYHIP_SW_TILE64=0: the 128 / 152-row tiles of rounds 1-5 -- every pair equal either way
    monkeypatch.setenv("POLYHIP_SW_TILE64", "0")
    got128, path128, _ = run(True)
    monkeypatch.delenv("POLYHIP_SW_TILE64", raising=False)
    assert path128 == 7
    for g, w in zip(got128, want):
        assert (g == w).all()
    assert int(got[0].max()) == 5 * maxA if maxA <= LB else int(got[0].max()) >= 4 * LB  # a read longer than the reference wraps around it
    om = orc.SubstitutionMatrix("-ACGT", "-ACGT", orc.NUC_4_SCORES)
    refb = ref.tobytes()
    for p in range(0, n, 1009):
        a = flat[offs[p]:offs[p + 1]].tobytes()
        s, _, _, ea_, eb_ = orc.smith_waterman(a, refb, om, -2)
        assert (int(got[0][p]), int(got[1][p]), int(got[2][p]), int(got[3][p])) == (s, ea_, eb_, 0), p


@pytest.mark.parametrize("L", [150, 250, 600])
def test_near_ties_are_located_without_the_full_sweep(al, monkeypatch, L):
    """Round 6: a maximum that several 4-column blocks reach is resolved by the locate step itself when the blocks lie within
    15 blocks of the first (sw_locate16_kernel up to 152 rows, sw_locate_kernel up to 256, the one-wave-per-pair locate
    kernels beyond).  The reference carries short tandem repeats (periods of 8..72 bp, a few more copies than the reads made
    from them), so reads align equally well at two or three shifts: spans of 2, 5, 7, 9, 10, 14, 15, 18 ... blocks -- on both
    sides of the limit -- next to ordinary mutated reads.  Score, endA, endB and err of every pair equal the exact 32-bit /
    full-sweep kernels (POLYHIP_SW_PACKED=0); a sample of the tie reads equals the oracle."""
    import torch
    align = al[0]
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(600 + L)
    LB = 6000
    ref = orc.synth_dna(0xC4, LB).copy()
    acgt = np.frombuffer(b"ACGT", np.uint8)
    regions = []
    at = 100
    for period in (8, 20, 28, 36, 40, 60, 64, 72):
        copies_read = max(2, (L - 10) // period)
        for extra in (1, 2):
            unit = acgt[rng.integers(0, 4, period)]  # (a unit of its own per region: the first shift is inside the region)
            ncopy = copies_read + extra
            if at + ncopy * period + 50 > LB:
                break
            ref[at:at + ncopy * period] = np.tile(unit, ncopy)
            regions.append((at, period, copies_read))
            at += ncopy * period + int(rng.integers(30, 90))
    assert len(regions) >= 6
    n = 50_001 if L <= 256 else 16_500  # enough pairs for the packed pass (path 3 / 7)
    reads, lens, is_tie = np.zeros((n, L), np.uint8), np.zeros(n, np.int64), np.zeros(n, bool)
    for p in range(n):
        if p % 3 == 0:  # a read made of whole periods: equally good at every shift the reference offers
            a0, period, k = regions[(p // 3) % len(regions)]
            r = ref[a0:a0 + k * period].copy()
            if p % 9 == 0 and len(r) > 4:  # ... some with a substitution (the same at every shift)
                r[int(rng.integers(0, len(r)))] = acgt[rng.integers(0, 4)]
            is_tie[p] = True
        else:
            a0 = int(rng.integers(0, LB - L))
            r = ref[a0:a0 + int(rng.integers(L // 2, L + 1))].copy()
            hit = rng.random(len(r)) < 0.06
            r[hit] = acgt[rng.integers(0, 4, int(hit.sum()))]
        lens[p] = len(r)
        reads[p, :len(r)] = r
    lens[0] = L if L <= LB else lens[0]
    reads[0, :L] = ref[3000:3000 + L]
    offs = np.zeros(n + 1, np.int64)
    offs[1:] = np.cumsum(lens)
    flat = np.concatenate([reads[i, :lens[i]] for i in range(n)])
    sc = _scoring(al, "-ACGT", al[2].NUC_4, -2)
    A, offA, B = torch.from_numpy(flat.copy()).to(dev), torch.from_numpy(offs).to(dev), torch.from_numpy(ref.copy()).to(dev)

    def run(packed):
        if packed:
            monkeypatch.delenv("POLYHIP_SW_PACKED", raising=False)
        else:
            monkeypatch.setenv("POLYHIP_SW_PACKED", "0")
        score = torch.full((n,), -7, dtype=torch.int64, device=dev)
        ea, eb, er = (torch.full((n,), -7, dtype=torch.int32, device=dev) for _ in range(3))
        work = torch.empty(align.sw_workspace_bytes(sc, n, L, LB), dtype=torch.uint8, device=dev)
        align.sw_batch_dev(sc, A, offA, L, B, None, LB, score, ea, eb, er, work)
        torch.cuda.synchronize()
        monkeypatch.delenv("POLYHIP_SW_PACKED", raising=False)
        return [t.cpu().numpy() for t in (score, ea, eb, er)], align.last_path()

    got, path = run(True)
    want, path0 = run(False)
    assert (path, path0) == ((3, 1) if L <= 256 else (7, 6))
    for g, w in zip(got, want):
        assert (g == w).all()
    om = orc.SubstitutionMatrix("-ACGT", "-ACGT", orc.NUC_4_SCORES)
    refb = ref.tobytes()
    ties = np.nonzero(is_tie)[0]
    for p in list(ties[:: max(1, len(ties) // 150)]) + [0, n - 1]:
        s, _, _, ea_, eb_ = orc.smith_waterman(flat[offs[p]:offs[p + 1]].tobytes(), refb, om, -2)
        assert (int(got[0][p]), int(got[1][p]), int(got[2][p]), int(got[3][p])) == (s, ea_, eb_, 0), p
    # the tie reads really are ties: their end column is the FIRST of the shifts the reference offers
    exact = [p for p in ties if p % 9 != 0][:200]
    for p in exact:
        a0, period, k = regions[(p // 3) % len(regions)]
        assert int(got[0][p]) == 5 * k * period and int(got[2][p]) == a0 + k * period, p
