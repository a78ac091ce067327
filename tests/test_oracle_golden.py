"""Pins the CPU oracle (oracle/) against every expectation the reference's own
tests hold for the hot path.  CPU only.

Each test names the reference test (file:line, relative to /root/reference)
whose assertion it restates.  If these fail, no GPU parity claim means
anything.
"""
import hashlib
import os

import numpy as np
import pytest

import oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _read(name):
    with open(os.path.join(GOLD, name)) as f:
        return f.read().strip()


# ---------------------------------------------------------------- murmur3 --
def test_murmur3_canonical_vectors():
    # The reference pins raw hashes only indirectly (mash_test.go:9-62), so the
    # hash itself is pinned on the canonical MurmurHash3_x86_32 seed-0 vectors.
    vec = {
        b"": 0x00000000,
        b"hello": 0x248BFA47,
        b"hello, world": 0x149BBB7F,
        b"19 Jan 2038 at 3:14:07 AM": 0xE31E8A70,
        b"The quick brown fox jumps over the lazy dog.": 0xD5C48BFC,
    }
    for k, v in vec.items():
        assert orc.murmur3_32(k) == v, k
    # seeded vectors from the MurmurHash3 SMHasher verification set
    assert orc.murmur3_32(b"", 1) == 0x514E28B7
    assert orc.murmur3_32(b"", 0xFFFFFFFF) == 0x81F16F39
    assert orc.murmur3_32(b"\x21\x43\x65\x87", 0x5082EDEE) == 0x2362F9DE
    assert orc.murmur3_32(b"\x21\x43\x65", 0) == 0x7E4A8634
    assert orc.murmur3_32(b"\x21\x43", 0) == 0xA0F7B07A
    assert orc.murmur3_32(b"\x21", 0) == 0x72661CF4


# ------------------------------------------------------------------- mash --
SEQ1 = "ATGCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGA"
SEQ2 = "ATCGATCGATCGATCGATCGATCGATCGATCGATCGAATGCGATCGATCGATCGATCGATCG"


def test_mash_TestMash():
    """search/mash/mash_test.go:9-62"""
    f1 = orc.Mash(17, 10)
    f1.Sketch(SEQ1)
    f2 = orc.Mash(17, 9)
    f2.Sketch(SEQ1)
    assert f1.Distance(f2) == 0  # :17-19
    assert f2.Distance(f1) == 0  # :21-24
    # pins duplicate-hash retention: sketch is 10x the same hash
    assert list(f1.Sketches) == [0x096698DE] * 10

    spoof = orc.Mash(17, 10)
    spoof.Sketches[0] = 0
    assert f1.Distance(spoof) == 1  # :26-32
    spoof = orc.Mash(17, 9)
    assert f1.Distance(spoof) == 1  # :34-39

    f1 = orc.Mash(17, 10)
    f1.Sketch(SEQ1)
    f2 = orc.Mash(17, 5)
    f2.Sketch(SEQ2)
    d = f1.Distance(f2)
    assert 0.19 < d < 0.21  # :47-50
    assert d == 0.19999999999999996  # the value the error message names
    assert list(f2.Sketches) == [0x08F7DC27] + [0x096698DE] * 4

    f1 = orc.Mash(17, 10)
    f1.Sketch(SEQ2)
    f2 = orc.Mash(17, 5)
    f2.Sketch(SEQ1)
    assert f1.Distance(f2) == 0  # :52-61


def test_mash_example():
    """search/mash/example_test.go:9-22 prints 0"""
    f1 = orc.Mash(17, 10)
    f1.Sketch(SEQ1)
    f2 = orc.Mash(17, 9)
    f2.Sketch(SEQ1)
    assert f1.Distance(f2) == 0


def test_mash_phix174_config1():
    """BASELINE config 1: mash.Sketch(data/phix174.gb, k=21, s=1000).

    Provisional goldens of SURVEY 8c (computed at survey time by an independent
    Python restatement); faithful (sort-on-accept) and insertion variants agree."""
    seq = _read("phix174.seq")
    assert len(seq) == 5386 and seq[:21] == "gagttttatcgcttccatgac"
    assert orc.murmur3_32(seq[:21]) == 0x1595F355
    for faithful in (False, True):
        m = orc.Mash(21, 1000)
        m.Sketch(seq, faithful=faithful)
        sk = m.Sketches
        assert [int(x) for x in sk[:5]] == [0x00006D6A, 0x0001273B, 0x002782B6, 0x002DCDAE, 0x004493CF]
        assert int(sk[499]) == 0x170B41DF and int(sk[998]) == 0x2F29F5F7 and int(sk[999]) == 0x2F30D748
        assert len(set(sk.tolist())) == 1000
        assert int(sk.astype(np.uint64).sum() & 0xFFFFFFFF) == 0x2DCF8AAA
        assert int(np.bitwise_xor.reduce(sk)) == 0x17EAB090
        assert hashlib.sha256(sk.astype("<u4").tobytes()).hexdigest() == \
            "943c9bb7559e8cb151b9382dbdab7ff8d1b642f64ad1ec7b9b03d709f8ad898a"


def test_tight_sketch_variant_equals_the_faithful_one_on_random_reads():
    """The full-size GPU parity tests compare EVERY row with the oracle's tight variant (faithful=0: an insertion instead
    of mash.go:84,97's sort.Slice; 3.6e8 k-mers/s on 16 cores against 6.6e6).  That is only a proof if the tight variant IS
    the faithful restatement: checked here on reads of the benchmark's generator, on reads with repeats and runs (equal
    hashes inside one sketch), on reads around the SketchSize boundary, for several (k, s)."""
    rng = np.random.default_rng(20)
    reads = [orc.synth_dna(0xC2, 40 * 10_000)[i * 10_000:(i + 1) * 10_000].tobytes() for i in range(40)]
    for _ in range(40):
        L = int(rng.integers(20, 4000))
        reads.append(bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8)))
    for unit_len, copies in ((7, 300), (37, 40), (200, 9), (1, 2500)):
        unit = bytes(rng.choice(list(b"ACGT"), unit_len).astype(np.uint8))
        reads.append(unit * copies)
        reads.append(bytes(rng.choice(list(b"ACGT"), 900).astype(np.uint8)) + unit * copies)
    for L in (1019, 1020, 1021, 1022, 1040):   # k = 21, s = 1000: the fill / first-sort / first-replacement boundary
        reads.append(bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8)))
    buf = np.frombuffer(b"".join(reads), np.uint8)
    offs = np.zeros(len(reads) + 1, np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    for k, s in ((21, 1000), (17, 200), (31, 64), (5, 1000)):
        prior = rng.integers(0, 1 << 32, (len(reads), s), dtype=np.uint32)   # Sketches survive where the reference leaves them
        a = orc.mash_sketch_batch(buf, offs, k, s, out=prior.copy(), faithful=True)
        b = orc.mash_sketch_batch(buf, offs, k, s, out=prior.copy(), faithful=False)
        assert (a == b).all(), (k, s, np.nonzero((a != b).any(axis=1))[0][:8])


def test_mash_sketch_quirks():
    """mash.go:68-104 behaviours a textbook MinHash would get wrong."""
    seq = orc.synth_dna(7, 400).tobytes()
    k = 21
    hashes = [orc.murmur3_32(seq[i:i + k]) for i in range(len(seq) - k + 1)]
    # (1) last k-mer skipped: n-k windows (mash.go:73)
    m = orc.Mash(k, 50)
    m.Sketch(seq)
    assert list(m.Sketches) == sorted(hashes[:-1])[:50]
    # (2) fewer windows than s: positional, unsorted, tail untouched
    m = orc.Mash(k, 1000)
    m.Sketches[:] = 0xABCD
    m.Sketch(seq)
    nwin = len(seq) - k
    assert list(m.Sketches[:nwin]) == hashes[:nwin]
    assert all(int(x) == 0xABCD for x in m.Sketches[nwin:])
    # (3) exactly s windows: sorted
    m = orc.Mash(k, nwin)
    m.Sketch(seq)
    assert list(m.Sketches) == sorted(hashes[:nwin])
    # (4) s-1 windows: positional (sort only happens at kmerStart == s-1)
    m = orc.Mash(k, nwin + 1)
    m.Sketch(seq)
    assert list(m.Sketches[:nwin]) == hashes[:nwin] and int(m.Sketches[nwin]) == 0
    # (5) shorter than k: no-op
    m = orc.Mash(k, 5)
    m.Sketch(seq[:k])
    assert not m.Sketches.any()
    m.Sketch(seq[:3])
    assert not m.Sketches.any()
    # (6) s == 0 panics as soon as there is a window; s == 1 when a smaller hash arrives
    with pytest.raises(orc.GoPanic):
        orc.Mash(k, 0).Sketch(seq)
    with pytest.raises(orc.GoPanic):
        orc.Mash(k, 1).Sketch(seq)


# ------------------------------------------------------------------ align --
def _mat3():
    sc = [[0, 0, 0, 0, 0], [0, 3, -3, -3, -3], [0, -3, 3, -3, -3], [0, -3, -3, 3, -3], [0, -3, -3, -3, 3]]
    return orc.SubstitutionMatrix("-ACGT", "-ACGT", sc)


def test_align_TestSmithWaterman():
    """search/align/align_test.go:139-292"""
    m = _mat3()
    assert orc.smith_waterman("TGTTACGG", "GGTTGACTA", m, -2)[:3] == (13, "GTT-AC", "GTTGAC")  # :158-175
    assert orc.smith_waterman("ACACACTA", "AGCACACA", m, -2)[:3] == (17, "A-CACACTA", "AGCACAC-A")  # :177-194
    assert orc.smith_waterman("", "GAT", m, -2)[:3] == (0, "", "")  # :199-215
    assert orc.smith_waterman("", "", m, -2)[:3] == (0, "", "")  # :217-234
    assert orc.smith_waterman("G", "A", m, -2)[:3] == (0, "", "")  # :236-253
    assert orc.smith_waterman("G", "G", m, -2)[:3] == (3, "G", "G")  # :255-272
    assert orc.smith_waterman("G", "GATTACA", m, -2)[:3] == (3, "G", "G")  # :274-291


def test_align_examples():
    """search/align/example_test.go:49-111"""
    pm1 = (2 * np.eye(5, dtype=int) - 1).tolist()
    m = orc.SubstitutionMatrix("ACGTU", "ACGTU", pm1)
    assert orc.smith_waterman("GATTACA", "GCATGCU", m, -1)[:3] == (2, "AT", "AT")  # :82
    assert orc.needleman_wunsch("GATTACA", "GCATGCU", m, -1) == (0, "G-ATTACA", "GCA-TGCU")  # :46
    # NUC_4 indexed by a mis-ordered alphabet {A,C,G,T,-}: 'A' hits the all-zero row
    m = orc.SubstitutionMatrix("ACGT-", "ACGT-", orc.NUC_4_SCORES)
    assert orc.smith_waterman("GATTACA", "GCATGCT", m, -1)[:3] == (15, "GATTAC", "GCATGC")  # :110


def test_align_TestNeedlemanWunsch():
    """search/align/align_test.go:11-137 (all eight score assertions)"""
    m = orc.SubstitutionMatrix("ACGTU", "ACGTU", (2 * np.eye(5, dtype=int) - 1).tolist())
    assert orc.needleman_wunsch("GATTACA", "GCATGCU", m, -1) == (0, "G-ATTACA", "GCA-TGCU")
    assert orc.needleman_wunsch("GATTACA", "GATTACA", m, -1) == (7, "GATTACA", "GATTACA")
    assert orc.needleman_wunsch("GATTACA", "GAT", m, -1)[0] == -1
    # traceback stops when either index hits 0 (align.go:141): leading residues dropped
    assert orc.needleman_wunsch("", "GAT", m, -1) == (-3, "", "")
    assert orc.needleman_wunsch("", "", m, -1) == (0, "", "")
    assert orc.needleman_wunsch("G", "A", m, -1)[0] == -1
    assert orc.needleman_wunsch("G", "G", m, -1) == (1, "G", "G")
    assert orc.needleman_wunsch("G", "GATTACA", m, -1)[0] == -5


def test_align_error_order():
    """align.go:189-191 + matrix.go:29-36: first failing cell in row-major order,
    first-alphabet error before second; no lookups when either string is empty."""
    m = _mat3()
    with pytest.raises(orc.AlphabetError, match="Symbol X not"):
        orc.smith_waterman("XG", "GY", m, -2)  # a[0] invalid: reported before b
    with pytest.raises(orc.AlphabetError, match="Symbol Y not"):
        orc.smith_waterman("GX", "GY", m, -2)  # a[0] ok -> first invalid b[j]
    with pytest.raises(orc.AlphabetError, match="Symbol X not"):
        orc.smith_waterman("GX", "GA", m, -2)
    assert orc.smith_waterman("", "!!", m, -2)[:3] == (0, "", "")
    assert orc.smith_waterman("!!", "", m, -2)[:3] == (0, "", "")


def test_matrix_score():
    """search/align/matrix/matrix_test.go:11-49"""
    m = orc.SubstitutionMatrix("-ACGT", "-ACGT", orc.NUC_4_SCORES)
    assert m.Score("A", "A") == 5 and m.Score("A", "C") == -4 and m.Score("T", "G") == -4
    assert m.Score("-", "A") == 0
    with pytest.raises(orc.AlphabetError):
        m.Score("X", "A")
    d = orc.DEFAULT_MATRIX
    assert d.Score("A", "A") == 1 and d.Score("A", "Z") == -1
    lut, va, vb = m.flatten()
    assert lut[ord("A"), ord("A")] == 5 and lut[ord("G"), ord("T")] == -4
    assert va[ord("A")] and not va[ord("N")] and not va[0xC3]


# ---------------------------------------------------------------- primers --
def test_primers_goldens():
    """primers/primers_test.go:13-84 and the full-precision values of SURVEY 8c"""
    assert orc.marmur_doty("ACGTCCGGACTT") == 31.0  # :24
    tm, dh, ds = orc.santalucia("ACGATGGCAGTAGCATGC", 0.1e-6, 350e-3, 0.0)
    assert abs(62.7 - tm) / 62.7 < 0.02  # :47
    assert abs(tm - 62.31695672635385) < 1e-9 and abs(dh - -144.0) < 1e-9
    assert abs(ds - -394.46768721086363) < 1e-9
    assert orc.reverse_complement("ACGTAGATCTACGT") == b"ACGTAGATCTACGT"  # :55-58
    tm, dh, ds = orc.santalucia("ACGTAGATCTACGT", 0.1e-6, 350e-3, 0.0)
    assert abs(47.428514 - tm) / 47.428514 < 0.02  # :62-63
    assert abs(tm - 47.42851359405711) < 1e-9
    assert abs(dh - -106.00000000000001) < 1e-9 and abs(ds - -298.6223490436017) < 1e-9
    tm = orc.melting_temp("GTAAAACGACGGCCAGT")
    assert abs(52.8 - tm) / 52.8 < 0.02  # :81
    assert abs(tm - 52.63382276100299) < 1e-9
    # lower case is folded (primers.go:71)
    assert orc.melting_temp("gtaaaacgacggccagt") == tm


def test_primers_go_log_matches_libm():
    import math
    rng = np.random.default_rng(1)
    for x in [50e-3, 350e-3, 500e-9 / 4, 0.1e-6 / 4, 0.1e-6, 1.0, 2.0, 0.5] + rng.uniform(1e-9, 10, 200).tolist():
        g, l = orc.go_log(x), math.log(x)
        assert g == l or abs(g - l) <= abs(math.ulp(l)), x


GENE = ("aataattacaccgagataacacatcatggataaaccgatactcaaagattctatgaagctatttgaggcacttggtacgatcaagtcgcgctcaatgtttggtggc"
        "ttcggacttttcgctgatgaaacgatgtttgcactggttgtgaatgatcaacttcacatacgagcagaccagcaaacttcatctaacttcgagaagcaagggcta"
        "aaaccgtacgtttataaaaagcgtggttttccagtcgttactaagtactacgcgatttccgacgacttgtgggaatccagtgaacgcttgatagaagtagcgaag"
        "aagtcgttagaacaagccaatttggaaaaaaagcaacaggcaagtagtaagcccgacaggttgaaagacctgcctaacttacgactagcgactgaacgaatgctt"
        "aagaaagctggtataaaatcagttgaacaacttgaagagaaaggtgcattgaatgcttacaaagcgatacgtgactctcactccgcaaaagtaagtattgagcta"
        "ctctgggctttagaaggagcgataaacggcacgcactggagcgtcgttcctcaatctcgcagagaagagctggaaaatgcgctttcttaa")


def _design_primers(seq: str, target: float):
    """primers/pcr/pcr.go:44-53 grow-until-Tm loop, on top of the oracle's MeltingTemp"""
    seq = seq.upper()
    fwd = seq[:15]
    add = 0
    while orc.melting_temp(fwd) < target:
        fwd = seq[: 15 + add]
        add += 1
    rev = orc.reverse_complement(seq[len(seq) - 15:]).decode()
    add = 0
    while orc.melting_temp(rev) < target:
        rev = orc.reverse_complement(seq[len(seq) - (15 + add):]).decode()
        add += 1
    return fwd, rev


def test_primers_pcr_threshold():
    """primers/pcr/example_test.go:54 -- pins the length at which Tm crosses 55.0"""
    assert _design_primers(GENE, 55.0) == ("AATAATTACACCGAGATAACACATCATGG", "TTAAGAAAGCGCATTTTCCAGC")


def test_pcr_ref_design_primers():
    """primers/pcr/example_test.go:36-55 through oracle/pcr_ref.py"""
    from oracle import pcr_ref
    assert pcr_ref.design_primers(GENE, 55.0) == ("AATAATTACACCGAGATAACACATCATGG", "TTAAGAAAGCGCATTTTCCAGC")
    assert pcr_ref.design_primers_with_overhangs(GENE, "TTATAGGTCTCATACT", "ATGAAGAGACCATATA", 55.0) == (
        "TTATAGGTCTCATACTAATAATTACACCGAGATAACACATCATGG", "TATATGGTCTCTTCATTTAAGAAAGCGCATTTTCCAGC")


PCR_FRAGMENT = ("TTATAGGTCTCATACT" + GENE.upper() + "ATGAAGAGACCATATA")


def test_pcr_ref_simulate_goldens():
    """primers/pcr/pcr_test.go:14-95, example_test.go:57-66"""
    from oracle import pcr_ref
    # ExampleSimulate / TestIssue279PCRBug: the amplicon is overhang + gene + overhang
    frags, err = pcr_ref.simulate([GENE], 55.0, False, ["TTATAGGTCTCATACTAATAATTACACCGAGATAACACATCATGG",
                                                      "TATATGGTCTCTTCATTTAAGAAAGCGCATTTTCCAGC"])
    assert err is None and frags == [PCR_FRAGMENT]
    frags, err = pcr_ref.simulate([GENE], 55.0, False, ["TATATGGTCTCTTCATTTAAGAAAGCGCATTTTCCAGC",
                                                      "TTATAGGTCTCATACTAATAATTACACCGAGATAACACATCATGG",
                                                      "CTGCAGGTCGACTCTAG"])
    assert err is None and len(frags) == 1 and frags[0] == PCR_FRAGMENT  # TestSimulatePrimerRejection, Issue279
    # TestSimulateMoreThanOneForward
    frags, _ = pcr_ref.simulate([GENE], 55.0, False, ["gatactcaaagattctatgaagctatttgaggcacttggtacg",
                                                    "tatcgctttgtaagcattcaatgcacctttctcttcaagttg",
                                                    "gtcgttcctcaatctcgcagagaagagctggaaaatg"])
    assert len(frags) == 1
    # TestSimulateCircular
    frags, _ = pcr_ref.simulate([GENE], 55.0, True, ["actctgggctttagaaggagcgataaacggc",
                                                   "aagtgcctcaaatagcttcatagaatctttgagtatcgg"])
    assert frags[0] == ("ACTCTGGGCTTTAGAAGGAGCGATAAACGGCACGCACTGGAGCGTCGTTCCTCAATCTCGCAGAGAAGAGCTGGAAAATGCGCTTTCTTAAAATAATTAC"
                        "ACCGAGATAACACATCATGGATAAACCGATACTCAAAGATTCTATGAAGCTATTTGAGGCACTT")
    # TestSimulateConcatemerization
    _, err = pcr_ref.simulate([GENE], 55.0, False, ["AATAATTACACCGAGATAACACATCATGG",
                                                  "CCATGATGTGTTATCTCGGTGTAATTATTTTAAGAAAGCGCATTTTCCAGC"])
    assert err == "Concatemerization detected in PCR."
    # pcr.go:174-178
    assert pcr_ref.simulate([GENE], 55.0, False, ["ACGT"]) == (None, "Primers are too short.")


# -------------------------------------------------------------- transform --
def test_transform_reverse_complement():
    """transform/transform_test.go:10-80, examples_test.go:9-30"""
    assert orc.reverse_complement("GATTACA") == b"TGTAATC"
    assert orc.reverse_complement("gattaca") == b"tgtaatc"
    assert orc.reverse_complement("ACGTN") == b"NACGT"
    assert orc.reverse_complement("AU-") == b"\x00\x00T"  # unmapped bytes -> 0x00 (transform.go:78-109)


# ---------------------------------------------------------------- seqhash --
def test_seqhash_TestHash():
    """seqhash/seqhash_test.go:12-66, example_test.go:11-31"""
    with pytest.raises(orc.SeqhashError, match="Only sequenceTypes"):
        orc.seqhash("ATGGGCTAA", "TNA", True, True)
    with pytest.raises(orc.SeqhashError, match="Got letter: X"):
        orc.seqhash("XTGGCCTAA", "DNA", True, True)
    with pytest.raises(orc.SeqhashError, match="Got letter: J"):
        orc.seqhash("MGCJ*", "PROTEIN", False, False)
    with pytest.raises(orc.SeqhashError, match="double stranded"):
        orc.seqhash("MGCS*", "PROTEIN", False, True)
    want = {
        ("TTAGCCCAT", "DNA", True, True): "v1_DCD_a376845b679740014f3eb501429b45e592ecc32a6ba8ba922cbe99217f6e9287",
        ("TTAGCCCAT", "DNA", True, False): "v1_DCS_ef79b6e62394e22a176942dfc6a5e62eeef7b5281ffcb2686ecde208ec836ba4",
        ("TTAGCCCAT", "DNA", False, True): "v1_DLD_c2c9fc44df72035082a152e94b04492182331bc3be2f62729d203e072211bdbf",
        ("TTAGCCCAT", "DNA", False, False): "v1_DLS_063ea37d1154351639f9a48546bdae62fd8a3c18f3d3d3061060c9a55352d967",
        ("TTAGCCCAT", "RNA", False, False): "v1_RLS_063ea37d1154351639f9a48546bdae62fd8a3c18f3d3d3061060c9a55352d967",
        ("MGC*", "PROTEIN", False, False): "v1_PLS_922ec11f5227ce77a42f07f565a7a1a479772b5cf3f1f6e93afc5ecbc0fd5955",
        ("ATGC", "DNA", False, True): "v1_DLD_f4028f93e08c5c23cbb8daa189b0a9802b378f1a1c919dcbcf1608a615f46350",
    }
    for args, h in want.items():
        assert orc.seqhash(*args) == h, args


def test_blake3_empty_and_tree_consistency():
    # published BLAKE3 digest of the empty input
    assert orc.blake3_256(b"").hex() == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    # multi-chunk inputs (> 1024 B: almost every real seqhash input): the BLAKE3 team's published test vectors
    # (test_vectors.json: input byte i = i % 251), first 32 bytes of the extended output, at the chunk / subtree
    # edges.  So the oracle's tree hashing is pinned on the published function, not only on the GPU's agreement.
    official = {
        1023: "10108970eeda3eb932baac1428c7a2163b0e924c9a9e25b35bba72b28f70bd11",
        1024: "42214739f095a406f3fc83deb889744ac00df831c10daa55189b5d121c855af7",
        1025: "d00278ae47eb27b34faecf67b4fe263f82d5412916c1ffd97c8cb7fb814b8444",
        2048: "e776b6028c7cd22a4d0ba182a8bf62205d2ef576467e838ed6f2529b85fba24a",
        2049: "5f4d72f40d7a5f82b15ca2b2e44b1de3c2ef86c426c95c1af0b6879522563030",
        3072: "b98cb0ff3623be03326b373de6b9095218513e64f1ee2edd2525c7ad1e5cffd2",
        4097: "9b4052b38f1c5fc8b1f9ff7ac7b27cd242487b3d890d15c96a1c25b8aa0fb995",
        8193: "bab6c09cb8ce8cf459261398d2e7aef35700bf488116ceb94a36d0f5f1b7bc3b",
    }
    for n, h in official.items():
        assert orc.blake3_256(bytes(i % 251 for i in range(n))).hex() == h, n
    a = bytes(i % 251 for i in range(5000))
    b = bytearray(a)
    b[4999] ^= 1
    assert orc.blake3_256(a) == orc.blake3_256(a)
    assert orc.blake3_256(a) != orc.blake3_256(bytes(b))


def test_seqhash_rotation():
    """seqhash/seqhash_test.go:68-91 (every rotation of pUC19), example_test.go:33-40"""
    assert orc.rotate_sequence("TTAGCCCAT") == b"AGCCCATTT"
    puc = _read("puc19.seq")
    assert len(puc) == 2686
    want = orc.rotate_sequence(puc)
    assert orc.booth_least_rotation(puc) == 2356
    assert want[:30] == b"aaaaaaaccaccgctaccagcggtggtttg"
    for r in range(0, len(puc), 1):
        assert orc.rotate_sequence(puc[r:] + puc[:r]) == want
    # clone/example_test.go:30-31: ExampleGoldenGate prints RotateSequence(construct) -- a reference-held
    # 3.7 kb least rotation: it rotates to itself, and so does every rotation of it
    gg = _read("clone_goldengate_rotated.seq")
    assert len(gg) == 3662 and orc.booth_least_rotation(gg) == 0 and orc.rotate_sequence(gg) == gg.encode()
    for r in range(1, len(gg), 37):
        assert orc.rotate_sequence(gg[r:] + gg[:r]) == gg.encode()
    # Booth == naive minimum over random strings, incl. periodic ones
    rng = np.random.default_rng(3)
    for _ in range(300):
        n = int(rng.integers(1, 40))
        s = bytes(rng.choice(list(b"ACGT"[: int(rng.integers(1, 5))]), n).tolist())
        assert orc.rotate_sequence(s) == min(s[i:] + s[:i] for i in range(n))


# --------------------------------------------------------------- synthetic --
def test_synth_dna_is_position_addressable():
    a = orc.synth_dna(0xC2, 10_000)
    assert set(a.tobytes()) <= set(b"ACGT")
    # same stream, shorter: prefix property the GPU generator relies on
    assert (orc.synth_dna(0xC2, 1000) == a[:1000]).all()
    counts = np.bincount(a, minlength=256)[[65, 67, 71, 84]]
    assert counts.min() > 2300


# ------------------------------------------------------------------ io/fastq --
def test_fastq_reference_fixtures():
    """io/fastq/fastq_test.go:59-66 (TestParseExceptions: all six files must fail), example_test.go:16-66
    (first identifier / sequence / quality of nanosavseq.fastq, ExampleParser's four identifiers).
    Fixtures: tests/golden/fastq/ = io/fastq/data/*.fastq."""
    from oracle import fastq_ref as fr
    d = os.path.join(GOLD, "fastq")
    recs, code, _ = fr.parse_all(open(os.path.join(d, "nanosavseq.fastq"), "rb").read())
    assert code == 0
    assert [r[0].decode() for r in recs] == ["e3cc70d5-90ef-49b6-bbe1-cfef99537d73", "92728f25-b658-426c-8cd7-d82dc70dbf71",
                                            "60907b6b-5e38-498e-9c07-f036ebd8c658", "990e110e-5e50-41a2-8ad5-92044d4465b8"]
    assert recs[0][1].startswith(b"GATGTGCGCCGTTCCAGTTGCGACGTACTATAATCCCCGGCAACACGGTGCTGATTC") and recs[0][1].endswith(b"CATGAGCAATACGTAACT")
    assert recs[0][2].startswith(b"$$&%&%#$)*59;/767C378411") and len(recs[0][2]) == len(recs[0][1])
    for name in ("noseq", "noquality", "noidentifier", "emptyseq", "noplus", "noquality2"):
        _, code, line = fr.parse_all(open(os.path.join(d, f"nanosavseq_{name}.fastq"), "rb").read())
        assert code != 0 and line > 0, name


# ------------------------------------------------------------------ io/fasta --
def test_fasta_reference_tests():
    """io/fasta/fasta_test.go:133-172 (TestParser), :206-215 (TestReadEmptyFasta), :233-241
    (TestParseEOFAfterName); example_test.go:18-36,100-114 on data/base.fasta (tests/golden/fasta/)."""
    from oracle import fasta_ref as fr
    assert fr.parse_all(b">humen\nGATTACA\nCATGAT") == ([], 0)                      # EOF-ended fasta not valid
    assert fr.parse_all(b">humen\nGATTACA\nCATGAT\n") == ([(b"humen", b"GATTACACATGAT")], 0)
    assert fr.parse_all(b">doggy or something\nGATTACA\n\nCATGAT\n>homunculus\nAAAA\n") == (
        [(b"doggy or something", b"GATTACACATGAT"), (b"homunculus", b"AAAA")], 0)
    recs, code = fr.parse_all(b"testing\natagtagtagtagtagatgatgatgatgagatg\n\n\n\n\n\n\n\n\n\n\n")
    assert recs == [] and code != 0
    recs, code = fr.parse_all(b">OK Fasta\nABGABA\n>NotOKFasta\n")
    assert code != 0
    recs, code = fr.parse_all(open(os.path.join(GOLD, "fasta", "base.fasta"), "rb").read())
    assert code == 0 and [r[0] for r in recs] == [b"gi|5524211|gb|AAD44166.1| cytochrome b [Elephas maximus maximus]",
                                                  b"MCHU - Calmodulin - Human, rabbit, bovine, rat, and chicken"]
    assert recs[1][1] == (b"ADQLTEEQIAEFKEAFSLFDKDGDGTITTKELGTVMRSLGQNPTEAELQDMINEVDADGNGTIDFPEFLTMMARKMKDTDSEEEIREAFRVFDKDGNGYISAAELRHVMTNLG"
                          b"EKLTDEEVDEMIREADIDGDGQVNYEEFVQMMTAK*")


# ------------------------------------------------- bench helpers of the oracle --
def test_oracle_loop_helpers_equal_the_per_call_functions():
    """orc_santalucia_scan / orc_mash_distance_matrix (bench.py's CPU baselines) are plain loops over the
    pinned per-call restatements"""
    g = bytes(orc.synth_dna(7, 300))
    tm, dh, ds = orc.santalucia_scan(g, 18, 30, 500e-9, 50e-3, 0.0)
    for L in (18, 25, 30):
        for i in (0, 17, len(g) - L):
            assert (tm[L - 18, i], dh[L - 18, i], ds[L - 18, i]) == orc.santalucia(g[i:i + L], 500e-9, 50e-3, 0.0)
        assert np.isnan(tm[L - 18, len(g) - L + 1:]).all()
    a, b = orc.Mash(17, 50), orc.Mash(17, 50)
    a.Sketch(g[:200])
    b.Sketch(g[100:])
    d = orc.mash_distance_matrix(np.stack([a.Sketches, b.Sketches]), np.stack([b.Sketches, a.Sketches]))
    assert d[0, 0] == a.Distance(b) and d[0, 1] == 0.0 and d[1, 0] == 0.0 and d[1, 1] == b.Distance(a)


# ---- clone's ligation + seqhash dedup (oracle/clone_ref.py; clone/clone.go:135-353) --------------------------
def _clone_parts():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "clone_parts.json")) as f:
        return json.load(f)


def test_clone_cut_with_enzyme_reference_cases():
    """clone/clone_test.go:18-140,216-228"""
    from oracle import clone_ref as cr
    bsai, comp = "GGTCTCAATGC", "ATGCAGAGACC"
    seq = "ATATATA" + comp + bsai + "ATGCATCGATCGACTAGCATG" + comp + bsai[:8]
    f = cr.cut_with_enzyme(seq, False, True, "BsaI")                                  # test(1)
    assert [x.Sequence for x in f] == ["ATGCATCGATCGACTAGCATG"]
    f = cr.cut_with_enzyme(seq, True, True, "BsaI")                                   # test(2)
    assert [x.Sequence for x in f] == ["ATGCATCGATCGACTAGCATG", "TATA"]
    seq = "ATATATATATATATAT" + bsai + "GCGCGCGCGCGCGCGCGCGC"
    f = cr.cut_with_enzyme(seq, False, False, "BsaI")                                 # test(3)
    assert [x.Sequence for x in f] == ["GCGCGCGCGCGCGCGCGCGC", "ATATATATATATATATGGTCTCA"]
    f = cr.cut_with_enzyme(seq, True, False, "BsaI")                                  # test(4)
    assert [x.Sequence for x in f] == ["GCGCGCGCGCGCGCGCGCGCATATATATATATATATGGTCTCA"]
    parts = _clone_parts()
    assert len(cr.cut_with_enzyme(parts["popen"][0], parts["popen"][1], False, "BbsI")) == 2   # test(5)
    seq = "AGCTGCTGTTTAAAGCTATTACTTTGAGACC"                                           # TestCutWithEnzymeRegression
    f = cr.cut_with_enzyme(seq, False, False, "BsaI")
    assert [(x.ForwardOverhang, x.ReverseOverhang) for x in f] == [("", "ACTT"), ("ACTT", "")]
    assert f[0].Sequence + f[0].ReverseOverhang + f[1].Sequence == seq
    p = parts["circular_cut_regression"][0]
    assert len(cr.cut_with_enzyme(p[0], p[1], True, "BsaI")) == 1                     # TestCircularCutRegression


def test_clone_ligation_reference_cases():
    """clone/clone_test.go:142-214 and clone/example_test.go:11-31: the restated recursion (with its per-call
    seqhash dedup and the sibling-persistent usedFragments list) gives the counts and the construct the
    reference's tests expect"""
    import os
    from oracle import clone_ref as cr
    o, i = cr.circular_ligate([cr.Fragment("AAAAAA", "GTTG", "CTAT"), cr.Fragment("AAAAAA", "CAAC", "ATAG")])
    assert (len(o), len(i)) == (1, 0)                                                 # TestCircularLigate
    parts = _clone_parts()
    o, i = cr.golden_gate([tuple(parts["popen"])] + [tuple(p) for p in parts["signal_killed"]], "BbsI")
    assert (len(o), len(i)) == (1, 4)                                                 # TestSignalKilledGoldenGate
    cr.golden_gate([tuple(parts["popen"])] + [tuple(p) for p in parts["panic"]], "BbsI")   # TestPanicGoldenGate: no panic
    o, i = cr.golden_gate([tuple(p) for p in parts["example_golden_gate"]], "BbsI")
    want = open(os.path.join(os.path.dirname(__file__), "golden", "clone_goldengate_rotated.seq")).read().strip()
    assert len(o) == 1 and orc.rotate_sequence(o[0]).decode() == want                 # ExampleGoldenGate
