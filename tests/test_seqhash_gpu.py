"""K5 parity: poly_amd.seqhash rotation (HIP, through the C ABI) vs the CPU oracle's
restatement of boothLeastRotation / RotateSequence (seqhash.go:78-138).  Index and
rotated bytes must be identical.

Mirrors seqhash/seqhash_test.go:68-91 (every rotation of pUC19 rotates to the same string)."""
import os

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def sh():
    from poly_amd import seqhash
    return seqhash


def _check(sh, seqs):
    from poly_amd.mash import _pack
    buf, offs = _pack(seqs)
    rot, out = sh.least_rotation_batch_packed(buf, offs, True)
    for i, s in enumerate(seqs):
        s = s if isinstance(s, bytes) else s.encode("latin-1")
        assert int(rot[i]) == orc.booth_least_rotation(s), (i, s[:40], len(s))
        assert out[int(offs[i]): int(offs[i + 1])].tobytes() == orc.rotate_sequence(s), (i, s[:40])


def test_reference_goldens(sh):
    assert sh.RotateSequence("TTAGCCCAT") == "AGCCCATTT"
    puc = open(os.path.join(GOLD, "puc19.seq")).read().strip()
    assert len(puc) == 2686
    r = sh.RotateSequence(puc)
    assert r.startswith("aaaaaaaccaccgctaccagcggtggtttg")
    assert r == orc.rotate_sequence(puc.encode()).decode()


def test_every_rotation_of_puc19(sh):
    """seqhash_test.go:68-91"""
    puc = open(os.path.join(GOLD, "puc19.seq")).read().strip()
    rots = [puc[i:] + puc[:i] for i in range(0, len(puc), 7)] + [puc[len(puc) - 1:] + puc[:len(puc) - 1]]
    out = sh.RotateBatch(rots)
    assert len(set(out)) == 1 and out[0] == sh.RotateSequence(puc)


def test_random_periodic_and_edge_cases(sh):
    rng = np.random.default_rng(21)
    seqs = [b"", b"A", b"AA", b"AB", b"BA", b"ABAB", b"BABA", b"AAB", b"ABA", b"BAA", b"ABABAA", b"ACGT", b"TGCA",
            b"A" * 1000, b"AC" * 700, b"ACG" * 333 + b"A", b"\xff\x00\xff\x00\x01", b"zyxwv" * 50,
            b"A" * 5000 + b"C", b"C" + b"A" * 5000, (b"ACGTTGCA" * 300)[3:] + (b"ACGTTGCA" * 300)[:3]]
    for _ in range(300):
        n = int(rng.integers(1, 400))
        al = [b"AB", b"ABC", b"ACGT", bytes(range(256))][int(rng.integers(0, 4))]
        if rng.random() < 0.5:
            p = int(rng.integers(1, max(2, n // 2)))
            base = bytes(rng.choice(list(al), p).astype(np.uint8))
            seqs.append((base * (n // p + 1))[:n])
        else:
            seqs.append(bytes(rng.choice(list(al), n).astype(np.uint8)))
    # plasmid-scale random DNA, a long low-complexity one, and one beyond the LDS staging limit
    seqs.append(bytes(orc.synth_dna(5, 10_000)))
    seqs.append(bytes(orc.synth_dna(6, 3000)) * 3)
    seqs.append(bytes(orc.synth_dna(7, 150_000)))
    seqs.append(b"AC" * 70_000 + b"A")
    _check(sh, seqs)


@pytest.mark.parametrize("wave_max", [None, "0", "100", "8192", "32768"])
def test_wave_and_workgroup_kernels_agree(sh, wave_max):
    """The wave-per-sequence kernel takes sequences up to 7 kB, the workgroup kernels what it marks (longer ones; beyond
    LDS from global memory).  POLYHIP_K5_WAVE_MAX moves the boundary: 0 = everything through the workgroup kernels,
    100 / 8192 / 32768 = other splits of the same batch.  Lengths on both sides of every boundary, > 1024 candidates of the
    least word (the list is full), tandem repeats that do and do not close, the least word wrapping around the origin."""
    rng = np.random.default_rng(5)
    seqs = [b"", b"G", b"CA"]
    for L in (7, 8, 9, 75, 76, 77, 99, 100, 101, 7167, 7168, 7169, 7192, 8191, 8192, 8193, 20000, 32767, 32768, 32769):
        seqs.append(bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8)))
    seqs.append(bytes(rng.choice(list(b"AC"), 8000).astype(np.uint8)) + b"CCCC")          # ~ 500 candidates, then rounds
    seqs.append(b"AAAC" * 1500)                                                          # 1500 candidates, closes on itself
    seqs.append((b"AAAC" * 1500)[:-1])                                                   # the same, not closing: list full
    seqs.append(b"GATTACA" * 700 + b"GAT")                                               # stalled rounds -> two-pointer search
    seqs.append(b"CGT" * 2000 + b"A" + b"CGT" * 30 + b"AA")                              # least word wraps: ...AA | CG...
    seqs.append(bytes(rng.choice(list(b"ACGT"), 124_000).astype(np.uint8)))               # beyond the LDS staging limit
    old = os.environ.pop("POLYHIP_K5_WAVE_MAX", None)
    try:
        if wave_max is not None:
            os.environ["POLYHIP_K5_WAVE_MAX"] = wave_max
        _check(sh, seqs)
    finally:
        os.environ.pop("POLYHIP_K5_WAVE_MAX", None)
        if old is not None:
            os.environ["POLYHIP_K5_WAVE_MAX"] = old


def test_device_batch(sh):
    import torch
    from poly_amd import mash
    dev = torch.device("cuda:0")
    n, L = 2000, 3000
    seqs = torch.empty(n * L, dtype=torch.uint8, device=dev)
    mash.synth_dna_dev(0x5EED, seqs)
    offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    rot = torch.zeros(n, dtype=torch.int64, device=dev)
    out = torch.zeros_like(seqs)
    sh.least_rotation_batch_dev(seqs, offs, L, rot, out)
    torch.cuda.synchronize()
    host = seqs.cpu().numpy()
    r = rot.cpu().numpy()
    o = out.cpu().numpy()
    for i in range(0, n, 37):
        s = host[i * L:(i + 1) * L].tobytes()
        assert int(r[i]) == orc.booth_least_rotation(s)
        assert o[i * L:(i + 1) * L].tobytes() == orc.rotate_sequence(s)


# ---- Hash (seqhash.go:141-224) ----------------------------------------------------------
def test_TestHash(sh):
    """seqhash/seqhash_test.go:12-66, example_test.go:11-31: error texts and the 7 exact seqhashes"""
    with pytest.raises(ValueError, match="Only sequenceTypes of DNA, RNA, or PROTEIN allowed. Got sequenceType: TNA"):
        sh.Hash("ATGGGCTAA", "TNA", True, True)
    with pytest.raises(ValueError, match="Only letters ATUGCYRSWKMBDHVNZ are allowed for DNA/RNA. Got letter: X"):
        sh.Hash("XTGGCCTAA", "DNA", True, True)
    with pytest.raises(ValueError, match=r"Only letters ACDEFGHIKLMNPQRSTVWYUO\*BXZ are allowed for Proteins. Got letter: J"):
        sh.Hash("MGCJ*", "PROTEIN", False, False)
    with pytest.raises(ValueError, match="Proteins cannot be double stranded"):
        sh.Hash("MGCS*", "PROTEIN", False, True)
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
        assert sh.Hash(*args) == h, args
    # published BLAKE3 digest of the empty input
    assert sh.Hash("", "DNA", False, False) == "v1_DLS_af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"


@pytest.mark.parametrize("stype,circular,ds", [("DNA", c, d) for c in (False, True) for d in (False, True)] +
                         [("RNA", True, True), ("RNA", False, False), ("PROTEIN", False, False), ("PROTEIN", True, False)])
def test_hash_batch_matches_oracle(sh, stype, circular, ds):
    rng = np.random.default_rng(hash((stype, circular, ds)) % (1 << 31))
    if stype == "PROTEIN":
        alpha = b"ACDEFGHIKLMNPQRSTVWYUO*BXZacdxz"
    else:
        alpha = b"ACGTacgtUuNRYSWKMBDHVZ"
    seqs = [b"", b"A", b"AT", b"TA", b"GAATTC", b"ACGU", b"acgu", b"ZZZ", b"AAAA"]
    # lengths around the 64-byte block and 1024-byte chunk edges, and multi-level trees
    for L in (63, 64, 65, 127, 128, 1023, 1024, 1025, 2047, 2048, 2049, 3072, 4097, 5000, 7 * 1024, 8 * 1024 + 1, 20_000,
              64 * 1024 - 1, 64 * 1024, 64 * 1024 + 1, 65 * 1024 + 7, 200_000):  # one thread merges up to 64 chunk values, a workgroup more
        seqs.append(bytes(rng.choice(list(alpha), L).astype(np.uint8)))
    for _ in range(60):
        seqs.append(bytes(rng.choice(list(alpha[:4] if stype != "PROTEIN" else alpha), int(rng.integers(1, 3000))).astype(np.uint8)))
    seqs += [b"ACGTXACGT", b"ACG-T", b"JJJ", b"acgtj"]  # alphabet errors (first offending letter)
    got = sh.HashBatch(seqs, stype, circular, ds)
    for s, g in zip(seqs, got):
        try:
            want = orc.seqhash(s, stype, circular, ds)
        except orc.SeqhashError as e:
            assert isinstance(g, ValueError) and str(g) == str(e), (s[:20], g, e)
            continue
        assert g == want, (s[:30], len(s))


@pytest.mark.parametrize("stype,ds", [("DNA", True), ("DNA", False), ("RNA", True), ("PROTEIN", False)])
def test_hash_batch_normalised_while_staged(sh, monkeypatch, stype, ds):
    """Round 6: a circular batch whose every sequence a K5 wave takes alone (<= 7168 bytes) is normalised BY K5 while it
    stages the bytes (no streaming pass in front).  Lower case, RNA's U, letters outside the alphabet in the first / a
    middle / the last position (the first offending letter is named), lengths 0, 1 and around the 16-byte pieces of the
    staging, the longest length a wave takes: equal to the oracle and to the pass it replaces (POLYHIP_S2_FOLD=0)."""
    rng = np.random.default_rng(hash((stype, ds)) % (1 << 31))
    alpha = b"ACDEFGHIKLMNPQRSTVWYUO*BXZacdxz" if stype == "PROTEIN" else b"ACGTacgtUuNRYSWKMBDHVZnryswkmbdhvz"
    seqs = [b"", b"A", b"a", b"u", b"AT", b"ta", b"GAATTC", b"acgu", b"ZZZ", b"AAAA", b"aAaA", b"J", b"j"]
    for L in list(range(2, 50)) + [63, 64, 65, 127, 1023, 1024, 1025, 2049, 4095, 4096, 5000, 7167, 7168]:
        seqs.append(bytes(rng.choice(list(alpha), L).astype(np.uint8)))
    for _ in range(200):
        seqs.append(bytes(rng.choice(list(alpha[:8]), int(rng.integers(1, 3000))).astype(np.uint8)))
    for L, at in ((40, 0), (40, 39), (40, 17), (5000, 0), (5000, 4999), (5000, 2500), (17, 16), (16, 15), (33, 32)):
        b = bytearray(rng.choice(list(alpha[:4]), L).astype(np.uint8))
        b[at] = ord("j")
        seqs.append(bytes(b))
        b[at // 2] = ord("-")  # two offending letters: the FIRST is named
        seqs.append(bytes(b))
    monkeypatch.delenv("POLYHIP_S2_FOLD", raising=False)
    got = sh.HashBatch(seqs, stype, True, ds)
    monkeypatch.setenv("POLYHIP_S2_FOLD", "0")
    unfolded = sh.HashBatch(seqs, stype, True, ds)
    monkeypatch.delenv("POLYHIP_S2_FOLD", raising=False)
    nerr = 0
    for s, g, u in zip(seqs, got, unfolded):
        assert str(g) == str(u), (s[:20], g, u)
        try:
            want = orc.seqhash(s, stype, True, ds)
        except orc.SeqhashError as e:
            assert isinstance(g, ValueError) and str(g) == str(e), (s[:20], g, e)
            nerr += 1
            continue
        assert g == want, (s[:30], len(s))
    assert nerr >= 18
    # ... and a batch WITHOUT any offending letter (the per-sequence pass behind K5 then returns at once)
    clean = [q for q, g in zip(seqs, got) if not isinstance(g, Exception)]
    got2 = sh.HashBatch(clean, stype, True, ds)
    assert got2 == [g for g in got if not isinstance(g, Exception)]


def test_runs_of_the_least_byte():
    """poly-A style inputs: long runs of the smallest byte (one candidate per run instead of one per position),
    runs that wrap around the origin, several runs of equal and different lengths, runs of a byte that is NOT
    the smallest, and sequences that are nothing but one byte -- index and rotated bytes equal Booth's."""
    from poly_amd import seqhash as sh
    rng = np.random.default_rng(77)
    seqs = []
    for _ in range(60):
        n = int(rng.integers(5, 3000))
        body = bytearray(rng.choice(list(b"ACGT"), n).astype(np.uint8))
        for _ in range(int(rng.integers(1, 4))):
            run = int(rng.integers(4, max(5, n // 2)))
            at = int(rng.integers(0, n))
            byte = int(rng.choice(list(b"AAAC")))  # mostly the least byte
            for t in range(run):
                body[(at + t) % n] = byte  # may wrap around the origin
        seqs.append(bytes(body))
    seqs += [b"A" * 7, b"A" * 1000, b"C" * 33, b"AAAAC", b"CAAAA", b"AACAA", b"AAAACAAAAC", b"AAAACAAAAAC" * 40,
             b"A" * 500 + b"C" + b"A" * 500, b"A" * 499 + b"C" + b"A" * 500, b"T" * 100 + b"A" * 100 + b"T" * 100 + b"A" * 100]
    offs = np.zeros(len(seqs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(q) for q in seqs])
    buf = np.frombuffer(b"".join(seqs), np.uint8).copy()
    rot, out = sh.least_rotation_batch_packed(buf, offs, True)
    for i, q in enumerate(seqs):
        assert int(rot[i]) == orc.booth_least_rotation(q), (i, q[:30], len(q))
        assert out[int(offs[i]): int(offs[i + 1])].tobytes() == orc.rotate_sequence(q), i


def test_clone_example_golden():
    """clone/example_test.go:30-31: the reference's ExampleGoldenGate prints RotateSequence of a 3.7 kb circular
    construct; that string is a reference-held least rotation -- it and every rotation of it rotate to it."""
    import os
    from poly_amd import seqhash as sh
    gg = open(os.path.join(os.path.dirname(__file__), "golden", "clone_goldengate_rotated.seq")).read().strip()
    assert sh.RotateSequence(gg) == gg
    rots = [gg[r:] + gg[:r] for r in range(0, len(gg), 11)]
    assert set(sh.RotateBatch(rots) if hasattr(sh, "RotateBatch") else [sh.RotateSequence(x) for x in rots]) == {gg}


def test_bytes_above_0x7f_are_refused_with_their_position():
    """seqhash.go:143 upper-cases the sequence as UTF-8: a byte >= 0x80 is (part of) a rune the alphabet check then names --
    the C ABI refuses such a batch (POLYHIP_ERR_INVALID with the byte and its offset; the Go overlay routes those sequences
    to the reference's own body).  Since round 4 the byte is found by the device's normalising pass, chunk by chunk, not by a
    host scan: first offending byte of the batch, also behind an earlier ASCII letter that is merely outside the alphabet,
    also on a device list, also in a later chunk of the host pipeline."""
    import os
    from poly_amd import _lib, devices, seqhash
    rng = np.random.default_rng(91)
    seqs = [bytes(rng.choice(list(b"ACGT"), int(L)).astype(np.uint8)) for L in rng.integers(50, 4000, 300)]
    seqs[20] = b"ACGTXACGT"                      # an alphabet error (per-sequence code), not a refusal
    buf, offs = __import__("poly_amd.mash", fromlist=["_pack"])._pack(seqs)
    hs, err = seqhash.seqhash_batch_packed(buf, offs, 0, True, True)
    assert err[20] == ((2 << 8) | ord("X")) and hs[19] == orc.seqhash(seqs[19], "DNA", True, True)
    bad = buf.copy()
    at = int(offs[200]) + 7
    bad[at] = 0xC3
    bad[int(offs[250]) + 1] = 0xFF               # a later one: the first is named
    for ids, chunk_mb in (([], None), ([0, 0, 0], None), ([], "1")):
        devices.set_devices(ids)
        if chunk_mb:
            os.environ["POLYHIP_HOST_CHUNK_MB"] = chunk_mb  # ~1 MB chunks: the byte sits in a later chunk
        try:
            with pytest.raises(_lib.PolyhipError) as e:
                seqhash.seqhash_batch_packed(bad, offs, 0, True, True)
        finally:
            os.environ.pop("POLYHIP_HOST_CHUNK_MB", None)
            devices.set_devices([])
        assert e.value.status == _lib.ERR_INVALID and f"byte 0xc3 at {at} is not ASCII" in e.value.message, e.value.message
