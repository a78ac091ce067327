"""NeedlemanWunsch parity: HIP through the C ABI vs the CPU oracle's restatement
of align.go:100-166 -- score and aligned strings, including the reference's quirk that the traceback
stops when either index reaches 0.

Mirrors search/align/align_test.go:11-137 and example_test.go:12-47."""
import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu
PM1 = (2 * np.eye(5, dtype=int) - 1).tolist()


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


def test_TestNeedlemanWunsch(al):
    """search/align/align_test.go:11-137: scores 0, 7, -1, -3, 0, -1, 1, -5; example_test.go:12-47"""
    align = al[0]
    sc = _scoring(al, "ACGTU", PM1, -1)
    om = orc.SubstitutionMatrix("ACGTU", "ACGTU", PM1)
    cases = [("GATTACA", "GCATGCU"), ("GATTACA", "GATTACA"), ("GATTACA", "GAT"), ("", "GAT"), ("", ""), ("G", "A"),
             ("G", "G"), ("G", "GATTACA"), ("GAT", "")]
    for a, b in cases:
        want = orc.needleman_wunsch(a.encode(), b.encode(), om, -1)
        got = align.NeedlemanWunsch(a, b, sc)
        ws = tuple(x.decode() if isinstance(x, bytes) else x for x in want[:3])
        assert got == ws, (a, b, got, ws)
    assert align.NeedlemanWunsch("GATTACA", "GCATGCU", sc) == (0, "G-ATTACA", "GCA-TGCU")  # example_test.go:46
    assert align.NeedlemanWunsch("", "GAT", sc) == (-3, "", "")
    with pytest.raises(al[1].Error, match="Symbol X not in alphabet"):
        align.NeedlemanWunsch("GAXTACA", "GCATGCU", sc)


@pytest.mark.parametrize("maxlen,generic", [(60, False), (150, False), (150, True), (250, False), (300, False), (300, True),
                                            (700, False), (1500, False), (3000, False), (4500, False)])
def test_batch_matches_oracle(al, monkeypatch, maxlen, generic):
    """ragged batches against the oracle: the register-tiled kernel (lenA <= 64 / 152 / 256 rows), the
    one-wave-per-pair kernel (257..4096) and the generic one (longer A, or POLYHIP_NW_GENERIC=1); invalid symbols;
    per-pair and shared B"""
    if generic:
        monkeypatch.setenv("POLYHIP_NW_GENERIC", "1")
    rng = np.random.default_rng(5)
    mat = [[0, 0, 0, 0, 0], [0, 3, -3, -3, -3], [0, -3, 3, -3, -3], [0, -3, -3, 3, -3], [0, -3, -3, -3, 3]]
    for gap in (-2, -1, 0, 1):
        sc = _scoring(al, "-ACGT", mat, gap)
        om = orc.SubstitutionMatrix("-ACGT", "-ACGT", mat)
        A = [bytes(rng.choice(list(b"ACGT"), int(rng.integers(0, min(maxlen, 160)))).astype(np.uint8)) for _ in range(300)]
        B = []
        for a in A:
            b = bytearray(a)
            for _ in range(int(rng.integers(0, 6))):
                if b and rng.random() < 0.5:
                    del b[int(rng.integers(0, len(b)))]
                else:
                    b.insert(int(rng.integers(0, len(b) + 1)), int(rng.choice(list(b"ACGT"))))
            B.append(bytes(b))
        long_a = bytes(rng.choice(list(b"ACGT"), maxlen).astype(np.uint8))
        long_b = bytearray(long_a)
        for _ in range(maxlen // 25):
            long_b[int(rng.integers(0, len(long_b)))] = int(rng.choice(list(b"ACGT")))
        del long_b[maxlen // 3: maxlen // 3 + 7]
        A += [(b"ACGT" * 1200)[:maxlen], b"A" * maxlen, b"ACNT", b"ACGT", b"TTTT", long_a, long_a[: maxlen // 2], long_a]
        B += [b"ACGA" * 65, b"A" * 17, b"ACGT", b"ACXT", b"", bytes(long_b), bytes(long_b), long_a[::-1]]
        pa, oa = _pack(A)
        pb, ob = _pack(B)
        score, err, sa, sb = al[0].nw_align_packed(sc, pa, oa, pb, ob)
        assert al[0].nw_last_path() == (2 if generic or maxlen > 4096 else 1 if maxlen <= 64 else 3)
        for p, (a, b) in enumerate(zip(A, B)):
            try:
                w = orc.needleman_wunsch(a, b, om, gap)
            except orc.AlphabetError as ex:
                sym = str(ex).split(" ")[1]
                assert int(err[p]) & 0xFF == ord(sym) and int(score[p]) == 0 and sa[p] == b"" and sb[p] == b""
                continue
            assert int(err[p]) == 0
            wa = w[1] if isinstance(w[1], bytes) else w[1].encode()
            wb = w[2] if isinstance(w[2], bytes) else w[2].encode()
            assert (int(score[p]), sa[p], sb[p]) == (w[0], wa, wb), (p, gap, a, b)
        # shared B
        score, err, sa, sb = al[0].nw_align_packed(sc, pa, oa, np.frombuffer(B[3], np.uint8), None)
        for p in range(0, len(A), 7):
            w = orc.needleman_wunsch(A[p], B[3], om, gap)
            wa = w[1] if isinstance(w[1], bytes) else w[1].encode()
            assert (int(score[p]), sa[p]) == (w[0], wa)
