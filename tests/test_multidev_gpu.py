"""One host-pointer (cgo) call spread over a device list (include/polyhip.h "one host call over several GPUs";
SURVEY 8b polyhip_init(n_devices), 8e's partitioning).  A list may name a device more than once, so the whole fan-out --
byte-balanced ragged splits, empty shards, the packed strings' two rounds, error selection and positions in messages,
scoring handles copied per device -- runs on this one-GPU box with [0, 0, 0] and friends.  Every entry point must give
what the one-device call gives AND what the oracle gives (search/mash/mash.go:68-140, search/align/align.go:100-232,
primers/primers.go:70-128, seqhash/seqhash.go:78-224)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LISTS = [[0, 0, 0], [0, 0], [0] * 7, [0]]


def _pack(seqs):
    offs = np.zeros(len(seqs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    return np.frombuffer(b"".join(seqs), np.uint8).copy(), offs


def _dna(rng, L):
    return bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8))


@pytest.fixture(autouse=True)
def _single_device_afterwards():
    from poly_amd import devices
    yield
    devices.set_devices([])


def test_device_list_roundtrip():
    from poly_amd import _lib, devices
    devices.set_devices([])  # (the suite itself may run under POLYHIP_DEVICES: see the module's last test)
    assert devices.get_devices() == []
    devices.set_devices([0, 0, 0])
    assert devices.get_devices() == [0, 0, 0]
    devices.init(1)
    assert devices.get_devices() == [0]
    devices.shutdown()
    assert devices.get_devices() == []
    n = _lib.lib().polyhip_device_count()
    with pytest.raises(_lib.PolyhipError) as e:
        devices.set_devices([0, n])
    assert "not one of" in str(e.value)
    assert devices.get_devices() == []  # a refused list changes nothing
    with devices.devices([0, 0]):
        assert devices.get_devices() == [0, 0]
    assert devices.get_devices() == []


# ---- K1 mash.Sketch ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("ids", LISTS)
def test_sketch_ragged_batch(ids):
    """reads of very different sizes (one of them most of the batch: the byte-balanced split leaves EMPTY shards), short
    reads whose rows keep prior state (mash.go:81-84), empty reads"""
    from poly_amd import devices, mash
    rng = np.random.default_rng(len(ids))
    k, s = 21, 200
    reads = [_dna(rng, int(L)) for L in rng.integers(0, 3000, 300)]
    reads[5] = _dna(rng, 400_000)
    reads[17] = b""
    reads[299] = _dna(rng, 30)
    buf, offs = _pack(reads)
    prior = rng.integers(0, 1 << 32, (len(reads), s), dtype=np.uint32)
    one = mash.sketch_batch_packed(buf, offs, k, s, out=prior.copy())
    want = orc.mash_sketch_batch(buf, offs, k, s, out=prior.copy())
    with devices.devices(ids):
        got = mash.sketch_batch_packed(buf, offs, k, s, out=prior.copy())
    assert (got == one).all() and (got == want).all()


def test_sketch_fewer_reads_than_devices():
    from poly_amd import devices, mash
    rng = np.random.default_rng(3)
    reads = [_dna(rng, 5000), _dna(rng, 100)]
    buf, offs = _pack(reads)
    want = orc.mash_sketch_batch(buf, offs, 17, 64)
    with devices.devices([0] * 5):
        got = mash.sketch_batch_packed(buf, offs, 17, 64)
        assert (got == want).all()
        assert mash.sketch_batch_packed(np.zeros(0, np.uint8), np.zeros(1, np.uint64), 17, 64).shape == (0, 64)


@pytest.mark.parametrize("s", [0, 1])
def test_sketch_size_below_two_names_the_globally_first_sequence(s):
    """mash.go:96,98 read by read: the panic names the FIRST panicking sequence of the whole batch (it sits in the last
    shard here), the rows of the others are written as the reference leaves them"""
    from poly_amd import _lib, devices, mash
    rng = np.random.default_rng(40 + s)
    k = 8
    reads = [_dna(rng, 8) for _ in range(40)]  # len == k: no window, no panic
    if s == 1:
        # window 0 fills Sketches[0]; a later window hashing below it panics: find such a read and a monotone one
        def panics(r):
            h = [orc.murmur3_32(r[i:i + k]) for i in range(len(r) - k)]
            return any(x < h[0] for x in h[1:])
        bad = next(r for r in (_dna(rng, 40) for _ in range(1000)) if panics(r))
        ok = next(r for r in (_dna(rng, 10) for _ in range(1000)) if not panics(r))
        reads[3] = ok
    else:
        bad = _dna(rng, 12)
    reads[37] = bad
    reads[39] = bad
    buf, offs = _pack(reads)
    for ids in ([], [0, 0, 0]):
        devices.set_devices(ids)
        out = np.full((len(reads), max(s, 1)), 7, np.uint32)[:, :s].copy()
        with pytest.raises(_lib.GoPanic) as e:
            mash.sketch_batch_packed(buf, offs, k, s, out=out)
        assert "on sequence 37 " in str(e.value), str(e.value)
        if s == 1:
            assert out[3, 0] == orc.murmur3_32(reads[3][:k])  # a sequence that does not panic has its row written


def test_sketch_offsets_error_position_is_global():
    from poly_amd import _lib, devices, mash
    rng = np.random.default_rng(5)
    reads = [_dna(rng, 500) for _ in range(60)]
    buf, offs = _pack(reads)
    offs[50] = offs[49] - 1
    msgs = []
    for ids in ([], [0, 0, 0]):
        devices.set_devices(ids)
        with pytest.raises(_lib.PolyhipError) as e:
            mash.sketch_batch_packed(buf, offs, 21, 10)
        msgs.append(e.value.message)
    assert msgs[0] == msgs[1] and "at 49" in msgs[0]


# ---- K2 distance matrix --------------------------------------------------------------------------------------------

@pytest.mark.parametrize("ids", [[0, 0, 0], [0] * 7])
def test_distance_matrix_rows_over_devices(ids):
    from poly_amd import devices, mash
    rng = np.random.default_rng(11)
    s = 128
    fam = [np.sort(rng.integers(0, 1 << 32, s, dtype=np.uint32)) for _ in range(10)]
    rows = []
    for i in range(230):
        b = fam[i % 10].copy()
        b[rng.integers(0, s, int(rng.integers(0, 40)))] = rng.integers(0, 1 << 32, 1, dtype=np.uint32)
        b.sort()
        rows.append(b)
    Y = np.stack(rows)
    X = Y[:101]
    one_c, one_d = mash.distance_matrix_packed(X, Y)
    with devices.devices(ids):
        got_c, got_d = mash.distance_matrix_packed(X, Y)
        few_c, _ = mash.distance_matrix_packed(Y[:2], Y, True, False)  # fewer rows than devices
    want = orc.mash_distance_matrix(X, Y)
    assert (got_c == one_c).all() and (got_d == one_d).all() and (got_d == want).all()
    assert (few_c == one_c[:2]).all()


@pytest.mark.parametrize("ids", [[], [0, 0, 0], [0] * 5])
def test_reads_to_distance_matrix_in_one_call(ids):
    """BASELINE configs[2] as one host call: reads -> sketches -> (peer exchange) -> row blocks.  Families of mutated
    genomes give non-trivial counts; a read shorter than k + s leaves a positional, stale-padded sketch (mash.go:81-84)
    whose pairs go through the reference's own merge; one read is most of the batch (empty shards)."""
    from poly_amd import devices, mash
    rng = np.random.default_rng(12)
    k, s = 21, 150
    genomes = [_dna(rng, 3000) for _ in range(6)]
    reads = []
    for i in range(90):
        g = bytearray(genomes[i % 6])
        for j in rng.integers(0, len(g), 25):
            g[int(j)] = int(rng.choice(list(b"ACGT")))
        reads.append(bytes(g))
    reads[10] = _dna(rng, 100)      # fewer than s windows
    reads[11] = b""
    reads[40] = genomes[0] * 40     # 120 kb: a shard of its own
    buf, offs = _pack(reads)
    prior = rng.integers(0, 1 << 32, (len(reads), s), dtype=np.uint32)
    want_sk = orc.mash_sketch_batch(buf, offs, k, s, out=prior.copy())
    want_c, want_d = mash.distance_matrix_packed(want_sk, want_sk)  # the two-call path, checked against the oracle below
    with devices.devices(ids):
        sk, c, d = mash.sketch_distance_matrix_packed(buf, offs, k, s, prior=prior.copy())
        sk0, c0, _ = mash.sketch_distance_matrix_packed(buf, offs, k, s, want_sketches=False, want_dist=False)
    assert (sk == want_sk).all() and (c == want_c).all() and (d == want_d).all()
    zero_prior = orc.mash_sketch_batch(buf, offs, k, s)
    assert sk0 is None and (c0 == mash.distance_matrix_packed(zero_prior, zero_prior, True, False)[0]).all()
    for i in range(0, 90, 7):
        for j in (0, 6, 10, 11, 40, 89):
            a, b = orc.Mash(k, s), orc.Mash(k, s)
            a.Sketches, b.Sketches = want_sk[i].copy(), want_sk[j].copy()
            assert d[i, j] == a.Distance(b)
    assert c.max() > 50  # the families do share hashes


def _family_reads(rng, nfam, copies, L, nsub):
    genomes = [_dna(rng, L) for _ in range(nfam)]
    reads = []
    for i in range(nfam * copies):
        g = bytearray(genomes[i % nfam])
        for j in rng.integers(0, len(g), nsub):
            g[int(j)] = int(rng.choice(list(b"ACGT")))
        reads.append(bytes(g))
    return reads


@pytest.mark.parametrize("ids", [[0, 0], [0, 0, 0], [0] * 5, [0] * 8])
def test_reads_to_distance_matrix_by_item_exchange(ids, monkeypatch):
    """The device list's ONE index without gathering the sketches (level 1 on a device's own rows, items exchanged by
    value range, level 2 on 1/N of the range, finished parts exchanged): every read has its s windows, so the exchange
    runs (last_path 1) -- equal to the gather (POLYHIP_K2_EXCHANGE=0, last_path 2), to the one-device call and to the
    two-call path on the oracle's sketches (mash.go:68-140).  Reads of very different sizes (ragged, empty shards), a
    tandem repeat (a hash many times in ONE sketch: occurrence numbers cross the exchange), both level-1 scatters."""
    from poly_amd import devices, mash
    rng = np.random.default_rng(len(ids))
    k, s = 21, 150
    reads = _family_reads(rng, 7, 30, 3000, 30)
    reads[3] = _dna(rng, 200) * 30            # period 200: every hash of the sketch 30 times over
    reads[50] = reads[3][:4000]
    reads[100] = _family_reads(rng, 1, 1, 150_000, 0)[0]  # most of the batch's bytes: shards without a read
    buf, offs = _pack(reads)
    want_sk = orc.mash_sketch_batch(buf, offs, k, s)
    want_c, want_d = mash.distance_matrix_packed(want_sk, want_sk)
    for mode in ({"POLYHIP_K2_STAGE": "1"}, {"POLYHIP_K2_STAGE": "0"}):  # both level-1 scatters of the two-level build the exchange runs
        monkeypatch.delenv("POLYHIP_K2_STAGE", raising=False)
        for k_, v_ in mode.items():
            monkeypatch.setenv(k_, v_)
        with devices.devices(ids):
            monkeypatch.delenv("POLYHIP_K2_EXCHANGE", raising=False)
            sk, c, d = mash.sketch_distance_matrix_packed(buf, offs, k, s)
            assert mash.sketch_distance_matrix_last_path() == 1
            # the call explains itself (round 5): an aliased list moves everything by LOCAL copies -- on distinct devices
            # the same counts appear under peer (xGMI) or staged (through the host) -- and the rounds were timed
            info = mash.sketch_distance_matrix_last_info()
            assert info["path"] == 1 and info["devices"] == len(ids)
            assert info["peer_copies"] == 0 and info["staged_copies"] == 0 and info["local_copies"] > 0 and info["bytes_local"] > 0
            assert info["ms_sketch"] > 0 and info["ms_index"] > 0 and info["ms_join"] > 0
            monkeypatch.setenv("POLYHIP_K2_EXCHANGE", "0")
            sk2, c2, d2 = mash.sketch_distance_matrix_packed(buf, offs, k, s)
            assert mash.sketch_distance_matrix_last_path() == 2
            info2 = mash.sketch_distance_matrix_last_info()
            assert info2["path"] == 2 and info2["ms_index"] == 0 and info2["local_copies"] >= len(ids) - 1
            # the gather moves every device the other devices' rows: (N - 1) x the sketch array, less the empty shards' nothing
            assert info2["bytes_local"] == (len(ids) - 1) * len(reads) * s * 4
        monkeypatch.delenv("POLYHIP_K2_EXCHANGE", raising=False)
        assert (sk == want_sk).all() and (c == want_c).all() and (d == want_d).all()
        assert (sk2 == want_sk).all() and (c2 == want_c).all() and (d2 == want_d).all()
    for i in range(0, len(reads), 11):
        for j in (0, 3, 50, 100, 7, 209):
            assert int(c[i, j]) == orc.mash_shared(want_sk[i], want_sk[j])
    assert c[3, 50] >= 90 and c.max() == s


def test_item_exchange_falls_back_where_the_merge_needs_raw_sketches():
    """one read with fewer than s windows (a positional sketch, mash.go:81-84): its pairs are the merge's, which reads both
    raw sketches -- the devices gather the sketches as before (last_path 2) and the matrix is the one-device call's"""
    from poly_amd import devices, mash
    rng = np.random.default_rng(77)
    k, s = 21, 150
    reads = _family_reads(rng, 4, 10, 2500, 20)
    reads[17] = _dna(rng, 120)
    buf, offs = _pack(reads)
    one = mash.sketch_distance_matrix_packed(buf, offs, k, s)
    assert mash.sketch_distance_matrix_last_path() == 0
    with devices.devices([0, 0, 0]):
        got = mash.sketch_distance_matrix_packed(buf, offs, k, s)
        assert mash.sketch_distance_matrix_last_path() == 2
    for a, b in zip(one, got):
        assert (a == b).all()


def test_item_exchange_with_compact_items_at_full_read_length():
    """12,000 reads of 10 kb (k = 21, s = 1000; 6,000 random ones and a mutated copy of each): 1.2e7 index items -- enough
    for the compact 4-byte item format (bucket shift 9), which the small tests do not reach -- over five aliased devices:
    counts equal to the gather's and the one-device call's on all 1.44e8 pairs, sampled pairs equal to the oracle's merge"""
    import torch
    from poly_amd import devices, mash
    n, L, k, s = 12_000, 10_000, 21, 1000
    d = torch.empty(n // 2 * L, dtype=torch.uint8, device=torch.device("cuda:0"))
    mash.synth_dna_dev(0xE4, d)
    half = d.cpu().numpy().reshape(n // 2, L)
    del d
    rng = np.random.default_rng(8)
    twin = half.copy()
    hit = rng.random(twin.shape) < 0.004
    twin[hit] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(hit.sum()))]
    host = np.concatenate([half, twin]).reshape(-1)
    offs = np.arange(0, (n + 1) * L, L, dtype=np.uint64)
    sk1, c1, _ = mash.sketch_distance_matrix_packed(host, offs, k, s, want_dist=False)
    with devices.devices([0] * 5):
        sk5, c5, _ = mash.sketch_distance_matrix_packed(host, offs, k, s, want_dist=False)
        assert mash.sketch_distance_matrix_last_path() == 1
    assert (sk5 == sk1).all() and (c5 == c1).all()
    assert (np.diagonal(c5) == s).all() and int((c5 != 0).sum()) >= 2 * n
    for i in rng.integers(0, n // 2, 40):
        for j in (int(i), int(i) + n // 2, int(rng.integers(0, n))):
            assert int(c5[i, j]) == orc.mash_shared(sk5[i], sk5[j]) and c5[i, j] == c5[j, i]
    assert c5[0, n // 2] > 300


def test_reads_to_distance_matrix_panics_like_the_reference():
    from poly_amd import _lib, devices, mash
    rng = np.random.default_rng(13)
    reads = [_dna(rng, 300) for _ in range(20)]
    buf, offs = _pack(reads)
    for ids in ([], [0, 0]):
        devices.set_devices(ids)
        with pytest.raises(_lib.GoPanic) as e:  # SketchSize 0 and a matrix: mash.go:117
            mash.sketch_distance_matrix_packed(buf, offs, 21, 0)
        assert "mash.go:117" in str(e.value)
        with pytest.raises(_lib.GoPanic) as e:  # SketchSize 0, sketches only: Sketch itself panics on the first read with a window
            mash.sketch_distance_matrix_packed(buf, offs, 21, 0, want_counts=False, want_dist=False)
        assert "on sequence 0 " in str(e.value)


# ---- K3 SmithWaterman / NeedlemanWunsch ----------------------------------------------------------------------------

def _nuc4(gap=-2):
    from poly_amd import align, alphabet, matrix
    a = alphabet.NewAlphabet(list("-ACGT"))
    return (align.NewScoring(matrix.NewSubstitutionMatrix(a, a, orc.NUC_4_SCORES), gap),
            orc.SubstitutionMatrix("-ACGT", "-ACGT", orc.NUC_4_SCORES), gap)


def _reads_from(rng, ref, n, lo, hi, bad_every=0):
    out = []
    for i in range(n):
        L = int(rng.integers(lo, hi))
        p = int(rng.integers(0, max(1, len(ref) - L)))
        r = bytearray(ref[p:p + L])
        for j in rng.integers(0, max(1, L), max(1, L // 15)):
            if L:
                r[int(j)] = int(rng.choice(list(b"ACGT")))
        if bad_every and i % bad_every == bad_every - 1 and L:
            r[int(rng.integers(0, L))] = ord("N")  # "Symbol N not in alphabet" (align.go:189-191)
        out.append(bytes(r))
    return out


def _oracle_sw(reads, refs, omat, gap):
    res = []
    for a, b in zip(reads, refs):
        try:
            res.append(orc.smith_waterman(a, b, omat, gap)[:3])
        except orc.AlphabetError as e:
            res.append(str(e))
    return res


@pytest.mark.parametrize("ids", [[0, 0, 0], [0] * 5])
def test_sw_score_and_strings_shared_reference(ids):
    from poly_amd import align, devices
    sc, omat, gap = _nuc4()
    rng = np.random.default_rng(21)
    ref = _dna(rng, 700)
    reads = _reads_from(rng, ref, 400, 0, 150, bad_every=37)
    reads[0] = b""
    A, offA = _pack(reads)
    B, _ = _pack([ref])
    one_s = align.sw_batch_packed(sc, A, offA, B)
    one_f = align.sw_align_packed(sc, A, offA, B)
    one_p = align.sw_align_strings_packed(sc, A, offA, B)
    with devices.devices(ids):
        got_s = align.sw_batch_packed(sc, A, offA, B)
        got_f = align.sw_align_packed(sc, A, offA, B)
        got_p = align.sw_align_strings_packed(sc, A, offA, B)
        tight = align.sw_align_strings_packed(sc, A, offA, B, capacity=10)  # too small: the retry path gets the exact size
    for g, o in zip(got_s, one_s):
        assert (g == o).all()
    for got, one in ((got_f, one_f), (got_p, one_p), (tight, one_p)):
        for q in range(4):
            assert (got[q] == one[q]).all()
        assert got[4] == one[4] and got[5] == one[5]
    want = _oracle_sw(reads, [ref] * len(reads), omat, gap)
    for p, w in enumerate(want):
        if isinstance(w, str):
            assert got_p[3][p] != 0 and chr(int(got_p[3][p]) & 0xFF) == w.split(" ")[1]
            assert got_p[4][p] == b"" and got_p[0][p] == 0
        else:
            assert (int(got_p[0][p]), got_p[4][p].decode(), got_p[5][p].decode()) == w


def test_sw_packed_strings_in_chunks_on_a_device_list(monkeypatch):
    """more pairs than one chunk of a shard (262,144): the running total stays on the device between chunks"""
    from poly_amd import align, devices
    sc, omat, gap = _nuc4()
    rng = np.random.default_rng(22)
    ref = _dna(rng, 120)
    reads = _reads_from(rng, ref, 600_000, 20, 40)
    A, offA = _pack(reads)
    B, _ = _pack([ref])
    one = align.sw_align_strings_packed(sc, A, offA, B)
    with devices.devices([0, 0]):
        got = align.sw_align_strings_packed(sc, A, offA, B)
    for q in range(4):
        assert (got[q] == one[q]).all()
    assert got[4] == one[4] and got[5] == one[5]
    for p in rng.integers(0, len(reads), 200):
        assert (int(got[0][p]), got[4][p].decode(), got[5][p].decode()) == orc.smith_waterman(reads[p], ref, omat, gap)[:3]


def test_sw_and_nw_per_pair_references():
    from poly_amd import align, devices
    sc, omat, gap = _nuc4(-3)
    rng = np.random.default_rng(23)
    refs = [_dna(rng, int(L)) for L in rng.integers(0, 90, 150)]
    reads = [_reads_from(rng, r, 1, 0, max(1, len(r)))[0] if len(r) > 4 else _dna(rng, 7) for r in refs]
    A, offA = _pack(reads)
    B, offB = _pack(refs)
    one_sw = align.sw_align_packed(sc, A, offA, B, offB)
    one_nw = align.nw_align_packed(sc, A, offA, B, offB)
    with devices.devices([0, 0, 0]):
        got_sw = align.sw_align_packed(sc, A, offA, B, offB)
        got_nw = align.nw_align_packed(sc, A, offA, B, offB)
    for q in range(4):
        assert (got_sw[q] == one_sw[q]).all()
    assert got_sw[4] == one_sw[4] and got_sw[5] == one_sw[5]
    assert (got_nw[0] == one_nw[0]).all() and (got_nw[1] == one_nw[1]).all() and got_nw[2:] == one_nw[2:]
    for p in range(len(reads)):
        assert (int(got_sw[0][p]), got_sw[4][p].decode(), got_sw[5][p].decode()) == orc.smith_waterman(reads[p], refs[p], omat, gap)[:3]
        assert (int(got_nw[0][p]), got_nw[2][p].decode(), got_nw[3][p].decode()) == orc.needleman_wunsch(reads[p], refs[p], omat, gap)


# ---- K4 primers ----------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("ids", [[0, 0, 0], [0] * 7])
def test_santalucia_scan_ranges_with_halo(ids):
    from poly_amd import devices, primers
    rng = np.random.default_rng(31)
    genome = bytearray(_dna(rng, 5000))
    genome[100:110] = b"acgtNNacgt"  # lower case folds, N contributes {0, 0} (primers.go:42-63,71)
    genome = bytes(genome)
    want = orc.santalucia_scan(genome, 18, 30, 500e-9, 50e-3, 0.0)
    one = primers.SantaLuciaScan(genome, 18, 30)
    one_first = primers.SantaLuciaScanFirst(genome, 18, 30, 55.0)
    with devices.devices(ids):
        got = primers.SantaLuciaScan(genome, 18, 30)
        got_first = primers.SantaLuciaScanFirst(genome, 18, 30, 55.0)
        tiny = primers.SantaLuciaScan(genome[:20], 18, 30)  # three starts over more devices
    for g, o, w in zip(got, one, want):
        assert np.array_equal(g, o, equal_nan=True) and np.array_equal(g, w, equal_nan=True)
    assert (got_first[0] == one_first[0]).all() and np.array_equal(got_first[1], one_first[1], equal_nan=True)
    assert np.array_equal(tiny[0], orc.santalucia_scan(genome[:20], 18, 30, 500e-9, 50e-3, 0.0)[0], equal_nan=True)


def test_primer_batches_and_error_positions():
    from poly_amd import _lib, devices, primers
    rng = np.random.default_rng(32)
    seqs = [_dna(rng, int(L)) for L in rng.integers(1, 60, 500)]
    buf, offs = _pack(seqs)
    one = primers.santalucia_batch_packed(buf, offs, 500e-9, 50e-3, 0.0)
    one_md = primers.marmurdoty_batch_packed(buf, offs)
    with devices.devices([0, 0, 0]):
        got = primers.santalucia_batch_packed(buf, offs, 500e-9, 50e-3, 0.0)
        got_md = primers.marmurdoty_batch_packed(buf, offs)
    for g, o in zip(got, one):
        assert (g == o).all()
    assert (got_md == one_md).all()
    for p in range(0, 500, 7):
        assert (got[0][p], got[1][p], got[2][p]) == orc.santalucia(seqs[p], 500e-9, 50e-3, 0.0)
        assert got_md[p] == orc.marmur_doty(seqs[p])
    # SantaLucia("") panics (primers.go:89): sequence 480 sits in the last shard, a non-ASCII byte in the middle one
    seqs[480] = b""
    buf2, offs2 = _pack(seqs)
    bad = buf.copy()
    bad[int(offs[250]) + 1] = 0xC3
    for ids in ([], [0, 0, 0]):
        devices.set_devices(ids)
        with pytest.raises(_lib.GoPanic) as e:
            primers.santalucia_batch_packed(buf2, offs2, 500e-9, 50e-3, 0.0)
        assert "sequence 480 is empty" in str(e.value)
        with pytest.raises(_lib.PolyhipError) as e:
            primers.marmurdoty_batch_packed(bad, offs)
        assert f"at {int(offs[250]) + 1} is not ASCII" in str(e.value)


# ---- K5 / S2 seqhash -----------------------------------------------------------------------------------------------

@pytest.mark.parametrize("ids", [[0, 0, 0], [0] * 6])
def test_least_rotation_and_seqhash(ids):
    from poly_amd import devices, seqhash
    rng = np.random.default_rng(41)
    seqs = [_dna(rng, int(L)) for L in rng.integers(0, 900, 240)]
    seqs[7] = b"ACGT" * 500          # periodic: any period start is a least rotation, the reference's index is the smallest
    seqs[100] = _dna(rng, 20_000)    # beyond the wave kernel
    seqs[200] = b"ACGTXACGT"         # seqhash.go:157 "Got letter: X"
    buf, offs = _pack(seqs)
    one_rot, one_out = seqhash.least_rotation_batch_packed(buf, offs, True)
    one_h = seqhash.seqhash_batch_packed(buf, offs, 0, True, True)
    with devices.devices(ids):
        got_rot, got_out = seqhash.least_rotation_batch_packed(buf, offs, True)
        got_h = seqhash.seqhash_batch_packed(buf, offs, 0, True, True)
    assert (got_rot == one_rot).all() and (got_out == one_out).all()
    assert got_h[0] == one_h[0] and (got_h[1] == one_h[1]).all()
    for p in list(range(0, 240, 9)) + [7, 100, 200]:
        assert int(got_rot[p]) == orc.booth_least_rotation(seqs[p])
        assert got_out[int(offs[p]):int(offs[p + 1])].tobytes() == orc.rotate_sequence(seqs[p])
        try:
            assert got_h[0][p] == orc.seqhash(seqs[p], "DNA", True, True)
        except orc.SeqhashError as e:
            assert got_h[0][p] == "" and chr(int(got_h[1][p]) & 0xFF) == str(e)[-1]


# ---- the environment variable, concurrent callers ------------------------------------------------------------------

def test_polyhip_devices_in_the_environment():
    """POLYHIP_DEVICES=0,0,0 is the list a process starts with; a malformed list is refused by polyhip_set_devices'
    parser and ignored at start-up"""
    code = (
        "import numpy as np, oracle as orc\n"
        "from poly_amd import devices, mash\n"
        "rng = np.random.default_rng(1)\n"
        "reads = [bytes(rng.choice(list(b'ACGT'), int(L)).astype(np.uint8)) for L in rng.integers(100, 4000, 50)]\n"
        "offs = np.zeros(51, np.uint64); offs[1:] = np.cumsum([len(r) for r in reads])\n"
        "buf = np.frombuffer(b''.join(reads), np.uint8).copy()\n"
        "got = mash.sketch_batch_packed(buf, offs, 21, 100)\n"
        "assert (got == orc.mash_sketch_batch(buf, offs, 21, 100)).all()\n"
        "print('devices', devices.get_devices())\n")
    for value, want in (("0,0,0", "devices [0, 0, 0]"), ("all", "devices [0]"), ("0;1", "devices []")):
        env = dict(os.environ, POLYHIP_DEVICES=value, PYTHONPATH=ROOT)
        res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert res.returncode == 0, res.stdout + res.stderr
        assert want in res.stdout, res.stdout + res.stderr


def test_two_caller_threads_on_two_aliased_devices():
    """tests/abi/abi_threads.c with a device list: cgo calls from two OS threads at once, every call fanned out over two
    workers that share the GPU; every result equal to the serial one-device run"""
    from poly_amd import build
    exe = build.build_abi_threads()
    env = dict(os.environ, POLYHIP_DEVICES="0,0")
    res = subprocess.run([exe, "2", "4"], env=env, capture_output=True, text=True, timeout=600)
    print(res.stdout, res.stderr)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "abi_threads ok: 2 threads" in res.stdout
    assert "device list: 2" in res.stdout


def test_other_suites_under_a_device_list():
    """The parity suites of the other modules -- every host-pointer call they make, their error cases, their
    polyhip_*_last_path assertions (the fan-out hands the first shard's kernel choice back to the caller's thread) -- once
    more in a process whose device list is 0,0,0 from the start.  (The whole `-m gpu` suite passes that way; this keeps
    the part of it that finishes in under a minute in the suite itself.)"""
    if os.environ.get("POLYHIP_DEVICES"):
        pytest.skip("already running under a device list")
    env = dict(os.environ, POLYHIP_DEVICES="0,0,0", PYTHONPATH=ROOT)
    suites = ["tests/test_align_gpu.py", "tests/test_primers_gpu.py", "tests/test_seqhash_gpu.py", "tests/test_stress_gpu.py",
              "tests/test_pcr_gpu.py", "tests/test_clone_gpu.py"]
    res = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu"] + suites, env=env, capture_output=True, text=True,
                         timeout=1500, cwd=ROOT)
    tail = (res.stdout + res.stderr)[-2000:]
    assert res.returncode == 0, tail
    assert " passed" in res.stdout and "failed" not in res.stdout, tail


def test_full_size_config4_over_a_device_list():
    """BASELINE configs[3] at full size through ONE host call on the list 0,0,0: all 1,000,000 reads of the contract's
    generator (windows of the 5 kb reference, 5 % substitutions + 1 % indels) against the shared reference, score + end +
    packed aligned strings.  Equal to the one-device call array by array (three shards of ~333k pairs, each more than one
    262,144-pair chunk: the strings' running total crosses chunks AND shards), 400 pairs against the oracle
    (align.go:171-232), and the size-independent properties on all of them: offsets ascending and consistent with the
    strings' total, both strings of a pair of one length, a score of 0 exactly where the strings are empty."""
    import torch
    from poly_amd import align, devices, workloads
    sc, omat, gap = _nuc4()
    n, LA, LB = 1_000_000, 150, 5000
    B_t, A_t = workloads.config4_reads(n, LA, LB, device=torch.device("cuda:0"))
    A = A_t.reshape(-1).cpu().numpy()
    B = B_t.cpu().numpy()
    del A_t, B_t
    offA = np.arange(0, (n + 1) * LA, LA, dtype=np.uint64)

    def call():
        score = np.zeros(n, np.int64)
        endA, endB, err = (np.zeros(n, np.uint32) for _ in range(3))
        off = np.zeros(n + 1, np.uint64)
        cap = n * 170
        alnA, alnB = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
        from poly_amd import _lib
        _lib.check(_lib.lib().polyhip_sw_align_batch_packed(
            sc.handle(), A.ctypes.data, offA.ctypes.data, n, B.ctypes.data, None, LB, score.ctypes.data, endA.ctypes.data,
            endB.ctypes.data, err.ctypes.data, alnA.ctypes.data, alnB.ctypes.data, off.ctypes.data, cap))
        return score, endA, endB, err, off, alnA, alnB

    one = call()
    with devices.devices([0, 0, 0]):
        got = call()
    total = int(one[4][n])
    for q in range(5):
        assert (got[q] == one[q]).all(), q
    assert (got[5][:total] == one[5][:total]).all() and (got[6][:total] == one[6][:total]).all()
    off = got[4].astype(np.int64)
    lens = np.diff(off)
    assert off[0] == 0 and (lens >= 0).all() and total == int(lens.sum()) and not got[3].any()
    assert ((lens == 0) == (got[0] == 0)).all() and lens.max() <= LA + LB
    rng = np.random.default_rng(4)
    ref = bytes(B)
    for p in rng.integers(0, n, 400):
        w = orc.smith_waterman(bytes(A[p * LA:(p + 1) * LA]), ref, omat, gap)
        assert (int(got[0][p]), got[5][off[p]:off[p + 1]].tobytes().decode(), got[6][off[p]:off[p + 1]].tobytes().decode(),
                int(got[1][p]), int(got[2][p])) == w, int(p)


def test_full_read_length_sketches_over_a_device_list():
    """BASELINE configs[1]'s read shape (10 kb, k = 21, s = 1000) through the host call on a device list: 60,000 reads
    (600 MB: three shards, each several 256 MB chunks of its own pipeline + the download helper), equal to the one-device
    call, 300 reads against the oracle (mash.go:68-104), every row ascending with no marker left behind."""
    import torch
    from poly_amd import devices, mash
    n, L, k, s = 60_000, 10_000, 21, 1000
    d = torch.empty(n * L, dtype=torch.uint8, device=torch.device("cuda:0"))
    mash.synth_dna_dev(0xC2, d)
    host = d.cpu().numpy()
    del d
    offs = np.arange(0, (n + 1) * L, L, dtype=np.uint64)
    one = mash.sketch_batch_packed(host, offs, k, s)
    with devices.devices([0, 0, 0]):
        got = mash.sketch_batch_packed(host, offs, k, s)
    assert (got == one).all()
    assert (np.diff(got.astype(np.int64), axis=1) >= 0).all()
    rng = np.random.default_rng(6)
    for i in rng.integers(0, n, 300):
        want = orc.mash_sketch_batch(host[i * L:(i + 1) * L], offs[:2], k, s)[0]
        assert (got[i] == want).all(), int(i)
