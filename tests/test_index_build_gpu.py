"""K2's index build, round 5: the SLICED build on 4-byte intermediate items (csrc/mash_distance.hip, struct Layout) against
the two-level build it replaces as the default and against the oracle's merge (mash.go:107-135).  The sketches are synthetic
ascending u32 arrays -- what the index takes; K1 is not on this path -- sized so that small inputs meet the device-side
conditions of the new build (2^16 <= largest hash < 2^30, 4 <= bucket shift <= 10, compact items), and shaped to reach each
of its special paths: both id groups (> 65,536 sketches), slices longer than a round's slots, a coarse bucket beyond level
2's registers, hashes repeated inside a sketch, irregular sketches, and the hand-over to the two-level build."""
import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mash():
    from poly_amd import mash
    return mash


@pytest.fixture(autouse=True)
def _clean_env(monkeypatch):
    for k in ("POLYHIP_K2_B4", "POLYHIP_K2_B4_SLOTS", "POLYHIP_K2_B4_TL", "POLYHIP_K2_STAGE", "POLYHIP_K2_COMPACT", "POLYHIP_K2_DENSE",
              "POLYHIP_K2_REGROW"):
        monkeypatch.delenv(k, raising=False)


def _families(rng, nfam, copies, s, bits, sub=0.1):
    """related ascending sketches with hashes below 2^bits (no hash twice in one sketch)"""
    out = np.empty((nfam * copies, s), np.uint32)
    for f in range(nfam):
        base = rng.choice(1 << bits, s, replace=False).astype(np.uint32)
        for c in range(copies):
            m = base.copy()
            hit = rng.random(s) < sub
            m[hit] = rng.integers(0, 1 << bits, int(hit.sum()), dtype=np.uint32)
            m = np.unique(m)
            while len(m) < s:  # a replacement hit a value the sketch holds already
                m = np.unique(np.concatenate([m, rng.integers(0, 1 << bits, s - len(m), dtype=np.uint32)]))
            out[f * copies + c] = m
    return out


def _counts(mash, X, Y, monkeypatch=None, env=None):
    import torch
    dev = torch.device("cuda:0")
    if env:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
    Xt = torch.from_numpy(X.view(np.int32)).to(dev)
    Yt = torch.from_numpy(Y.view(np.int32)).to(dev)
    ct = torch.full((len(X), len(Y)), -1, dtype=torch.int16, device=dev)
    work = torch.zeros(mash.shared_counts_workspace_bytes(len(X), X.shape[1], len(Y), Y.shape[1]), dtype=torch.uint8, device=dev)
    mash.shared_counts_dev(Xt, Yt, ct, work)
    torch.cuda.synchronize()
    info = mash.index_build_info(work)
    info["item_bytes"] = mash.index_item_bytes(work)
    if env:
        for k in env:
            monkeypatch.delenv(k)
    return ct.cpu().numpy().view(np.uint16), info


def _check_rows(counts, X, Y, rows):
    for i in rows:
        for j in range(len(Y)):
            assert int(counts[i, j]) == orc.mash_shared(X[i], Y[j]), (i, j)


@pytest.mark.parametrize("slots", ["128", "64", "32"])
def test_sliced_build_equals_two_level_build_and_oracle(mash, monkeypatch, slots):
    rng = np.random.default_rng(501)
    S = _families(rng, 60, 50, 500, 26)   # 3000 sketches x 500: 2^17 buckets, shift 9, 1024 coarse buckets
    monkeypatch.setenv("POLYHIP_K2_B4_SLOTS", slots)
    got, info = _counts(mash, S, S)
    assert info["build"] == 1 and info["item_bytes"] == 4 and info["two_pass_buckets"] == 0 and info["repeated"] == 0
    assert info["coarse"] == 1024 and info["parts"] >= 2
    old, oinfo = _counts(mash, S, S, monkeypatch, {"POLYHIP_K2_B4": "0"})
    assert oinfo["build"] == 0 and oinfo["item_bytes"] == 4
    assert (got == old).all()
    assert (got.diagonal() == 500).all()
    _check_rows(got, S, S, range(0, len(S), 211))


def test_geometry_follows_the_slice_length(mash, monkeypatch):
    """POLYHIP_K2_B4_TL (hashes of a sketch per part, the tuning knob): more parts, the same counts"""
    rng = np.random.default_rng(502)
    S = _families(rng, 20, 30, 1000, 26)   # 600 x 1000: 2^16 buckets, shift 10
    ref, info = _counts(mash, S, S)
    assert info["build"] == 1
    for tl in ("16", "40", "250", "1024"):
        got, i2 = _counts(mash, S, S, monkeypatch, {"POLYHIP_K2_B4_TL": tl})
        assert i2["build"] == 1 and (got == ref).all(), tl
        assert i2["parts"] * i2["coarse_per_part"] >= (int(S.max()) >> 16) + 1
    _check_rows(ref, S, S, (0, 299, 599))


def test_both_id_groups(mash, monkeypatch):
    """more than 65,536 sketches: an intermediate item carries 16 id bits, the 17th is which segment of its coarse bucket
    it lies in; related sketches sit on both sides of the boundary"""
    rng = np.random.default_rng(503)
    fam = _families(rng, 700, 50, 64, 26)             # 35,000
    Y = np.concatenate([fam, fam[::-1].copy()])       # 70,000: sketch j and 69,999 - j are equal
    Y[40_000:, 0] = 0                                 # ... but for their first hash in the upper half's tail
    Y.sort(axis=1)
    rows = np.r_[0:8, 34_990:35_010, 65_530:65_545, 69_990:70_000]
    X = Y[rows].copy()
    got, info = _counts(mash, X, Y)
    assert info["build"] == 1 and info["item_bytes"] == 4
    old, oinfo = _counts(mash, X, Y, monkeypatch, {"POLYHIP_K2_B4": "0"})
    assert oinfo["build"] == 0 and (got == old).all()
    for a, i in enumerate(rows):
        assert got[a, i] == 64
        assert got[a, 69_999 - i] >= 63
    cols = rng.choice(len(Y), 300, replace=False)
    for a in range(0, len(rows), 5):
        for j in list(cols) + [int(rows[a]), 69_999 - int(rows[a])]:
            assert int(got[a, j]) == orc.mash_shared(X[a], Y[j])


def test_slices_longer_than_a_round_and_sketches_of_other_scales(mash, monkeypatch):
    """sketches of long sequences crowd the low values: all their hashes lie in the first part of the value range, many
    rounds of slots; short ones spread over all parts"""
    rng = np.random.default_rng(504)
    wide = _families(rng, 10, 40, 400, 25)             # 1200 x 400: 2^15 buckets, shift 10, 512 coarse buckets
    narrow = _families(rng, 10, 40, 400, 19)           # 400 hashes below 2^19: the first part holds them all
    mid = _families(rng, 10, 40, 400, 22)
    S = np.concatenate([wide, narrow, mid])
    S = S[rng.permutation(len(S))]
    for slots in ("128", "64", "32"):
        got, info = _counts(mash, S, S, monkeypatch, {"POLYHIP_K2_B4_SLOTS": slots})
        assert info["build"] == 1
        old, _ = _counts(mash, S, S, monkeypatch, {"POLYHIP_K2_B4": "0"})
        assert (got == old).all()
    _check_rows(got, S, S, range(0, len(S), 97))


def test_a_coarse_bucket_beyond_the_registers(mash, monkeypatch):
    """40,000 sketches that share 40 hashes of one coarse bucket: 1.6M items in it -- level 2's two-pass path"""
    rng = np.random.default_rng(505)
    n, s = 40_000, 64
    shared = (np.uint32(0x01230000) + rng.choice(65536, 40, replace=False)).astype(np.uint32)
    Y = np.empty((n, s), np.uint32)
    Y[:, :40] = shared
    Y[:, 40:] = rng.integers(0, 1 << 26, (n, s - 40), dtype=np.uint32) | np.uint32(1 << 25)  # (never the shared bucket)
    Y.sort(axis=1)
    X = Y[:24].copy()
    X[3, :20] = np.arange(20, dtype=np.uint32)
    X.sort(axis=1)
    got, info = _counts(mash, X, Y)
    assert info["build"] == 1 and info["two_pass_buckets"] >= 1
    old, _ = _counts(mash, X, Y, monkeypatch, {"POLYHIP_K2_B4": "0"})
    assert (got == old).all()
    assert (got >= 20).all()
    cols = rng.choice(n, 200, replace=False)
    for i in (0, 3, 23):
        for j in cols:
            assert int(got[i, j]) == orc.mash_shared(X[i], Y[j])


def test_repeated_hashes_get_their_numbers(mash, monkeypatch):
    """copies of a hash inside ONE sketch (multiset semantics of the merge, mash.go:121-131): the check pass logs them,
    level 2 numbers them -- in ordinary buckets, in a bucket that takes two passes, at both ends of a sketch"""
    rng = np.random.default_rng(506)
    S = _families(rng, 30, 40, 300, 23)       # shift 8: a compact item numbers up to 7 copies
    n = len(S)
    for q in range(0, n, 7):                    # a pair of equal hashes somewhere
        e = int(rng.integers(1, 299))
        S[q, e] = S[q, e - 1]
    for q in range(3, n, 40):                   # runs of 2..5 copies, the same value in several sketches of a family
        f = q // 40 * 40
        v = S[f, 150]
        for d in range(6):
            c = 2 + (d % 4)
            S[f + d, 100:100 + c] = v
    S[5, :4] = S[5, 0]                          # at the front
    S[6, -3:] = S[6, -1]                        # at the back
    S.sort(axis=1)
    got, info = _counts(mash, S, S)
    assert info["build"] == 1 and info["item_bytes"] == 4 and info["repeated"] > n // 7
    old, oinfo = _counts(mash, S, S, monkeypatch, {"POLYHIP_K2_B4": "0"})
    assert oinfo["build"] == 0 and (got == old).all()
    _check_rows(got, S, S, list(range(0, 12)) + list(range(40, 50)) + [n - 1])
    # ... and in a coarse bucket beyond the registers
    n2, s2 = 36_000, 64
    Y = np.empty((n2, s2), np.uint32)
    Y[:, :32] = (np.uint32(0x00770000) + rng.choice(65536, 32, replace=False)).astype(np.uint32)
    Y[:, 32:] = rng.integers(0, 1 << 26, (n2, s2 - 32), dtype=np.uint32) | np.uint32(1 << 25)
    Y[::9, 1] = Y[::9, 0]
    Y[::31, 2] = Y[::31, 0]
    Y[::31, 1] = Y[::31, 0]
    Y.sort(axis=1)
    X = Y[:40].copy()
    got2, info2 = _counts(mash, X, Y)
    assert info2["build"] == 1 and info2["two_pass_buckets"] >= 1 and info2["repeated"] > 0
    old2, _ = _counts(mash, X, Y, monkeypatch, {"POLYHIP_K2_B4": "0"})
    assert (got2 == old2).all()
    cols = np.r_[0:64, rng.choice(n2, 100, replace=False)]
    for i in (0, 9, 31):
        for j in cols:
            assert int(got2[i, j]) == orc.mash_shared(X[i], Y[j])


def test_irregular_sketches_stay_out_of_the_index(mash, monkeypatch):
    rng = np.random.default_rng(507)
    S = _families(rng, 12, 25, 256, 22)       # 300 x 256: 2^13 buckets, shift 9
    S[17] = S[17][::-1].copy()                  # descending
    S[100, 10], S[100, 200] = S[100, 200], S[100, 10]
    S[205, :] = 0
    S[205, 0] = 5                               # positional / zero-padded (mash.go:81-84)
    got, info = _counts(mash, S, S)
    assert info["build"] == 1
    want = np.array([[orc.mash_shared(S[i], S[j]) for j in range(len(S))] for i in (0, 17, 100, 205, 299)], np.uint16)
    assert (got[[0, 17, 100, 205, 299]] == want).all()
    old, _ = _counts(mash, S, S, monkeypatch, {"POLYHIP_K2_B4": "0"})
    assert (got == old).all()


def test_hand_over_to_the_two_level_build(mash, monkeypatch):
    """decided on the device: a largest hash of 2^30 or more (plan: build 0), a sketch that repeats a hash more often than a
    compact item numbers (planned, then called off: build 2, 8-byte items)"""
    rng = np.random.default_rng(508)
    S = _families(rng, 20, 30, 500, 25)       # 600 x 500: 2^15 buckets, shift 10
    big = S.copy()
    big[11, -1] = np.uint32(0xC0000000)
    got, info = _counts(mash, big, big)
    assert info["build"] == 0
    _check_rows(got, big, big, (0, 11, 599))
    rep = S.copy()
    rep[4, :300] = rep[4, 0]
    rep[9, :100] = rep[4, 0]
    rep.sort(axis=1)
    got, info = _counts(mash, rep, rep)
    assert info["build"] == 2 and info["item_bytes"] == 8
    _check_rows(got, rep, rep, (0, 4, 9, 599))
    # more repeated hashes than the build's list of them holds (65,536 records): called off too, but the items stay compact
    many = np.sort(rng.integers(0, 1 << 26, (70_000, 64), dtype=np.uint32), axis=1)
    many[:, 1] = many[:, 0]                     # every sketch repeats its smallest hash once
    Xm = many[::3500].copy()
    got, info = _counts(mash, Xm, many)
    assert info["build"] == 2 and info["item_bytes"] == 4 and info["repeated"] >= 70_000
    cols = rng.choice(len(many), 150, replace=False)
    for a in range(0, len(Xm), 4):
        for j in list(cols) + [a * 3500]:
            assert int(got[a, j]) == orc.mash_shared(Xm[a], many[j])
    # the parts API keeps the two-level build's coarse buckets
    import torch
    dev = torch.device("cuda:0")
    St = torch.from_numpy(S.view(np.int32)).to(dev)
    work = torch.zeros(mash.shared_counts_workspace_bytes(len(S), 500, len(S), 500), dtype=torch.uint8, device=dev)
    mash.index_build_part_dev(St, 0, 1, work)
    torch.cuda.synchronize()
    assert mash.index_build_info(work)["build"] == 0
    mash.index_build_dev(St, work)
    torch.cuda.synchronize()
    assert mash.index_build_info(work)["build"] == 1
    ct = torch.zeros((len(S), len(S)), dtype=torch.int16, device=dev)
    mash.shared_counts_reuse_dev(St, St, ct, work)
    torch.cuda.synchronize()
    _check_rows(ct.cpu().numpy().view(np.uint16), S, S, (0, 300, 599))
