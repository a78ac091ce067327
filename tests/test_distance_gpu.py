"""K2 parity: poly_amd.mash Similarity / Distance / distance matrix (HIP, through
the C ABI) vs the CPU oracle's restatement of mash.go:107-140.  Counts are
integers and must be identical; distances are one fp64 divide + subtract and must
be bit-identical.

Mirrors search/mash/mash_test.go:9-62 and example_test.go where the reference has a test."""
import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu

S62 = "ATGCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGA"
S62B = "ATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGAT"[:62]


@pytest.fixture(scope="module")
def mash():
    from poly_amd import mash
    return mash


@pytest.fixture(autouse=True, params=["dense", "sparse", "wide", "staged", "twolevel", "zahead"])
def join_kind(request, monkeypatch):
    """every test runs twice: with the dense join in front (a counter per column in LDS; the default up to two
    stripes of columns; index built with the LDS-staged level-1 scatter) and with POLYHIP_K2_DENSE=0 (sparse LDS hash
    join, dense join only for overflowing rows) + POLYHIP_K2_STAGE=0 (the index's direct level-1 scatter)"""
    monkeypatch.delenv("POLYHIP_K2_DENSE", raising=False)
    monkeypatch.delenv("POLYHIP_K2_STAGE", raising=False)
    monkeypatch.delenv("POLYHIP_K2_COMPACT", raising=False)
    monkeypatch.delenv("POLYHIP_K2_REGROW", raising=False)
    monkeypatch.delenv("POLYHIP_K2_B4", raising=False)
    monkeypatch.delenv("POLYHIP_K2_ZAHEAD", raising=False)
    if request.param == "zahead":   # the dense join writing a row's zeros ahead, during the walk before (measured, not faster: opt-in)
        monkeypatch.setenv("POLYHIP_K2_ZAHEAD", "1")
    if request.param == "twolevel":   # the two-level index build on 8-byte intermediate items where the sliced build (round 5) is the default
        monkeypatch.setenv("POLYHIP_K2_B4", "0")
    if request.param == "staged":   # the dense join with the row staged in LDS (the default keeps a row of <= 1024 hashes in registers)
        monkeypatch.setenv("POLYHIP_K2_REGROW", "0")
    if request.param == "sparse":
        monkeypatch.setenv("POLYHIP_K2_DENSE", "0")
        monkeypatch.setenv("POLYHIP_K2_STAGE", "0")
    elif request.param == "wide":   # the dense join on 8-byte items (the default picks compact 4-byte items where they fit)
        monkeypatch.setenv("POLYHIP_K2_COMPACT", "0")
    return request.param


def _oracle_counts(X, Y):
    out = np.zeros((len(X), len(Y)), np.uint16)
    for i, x in enumerate(X):
        for j, y in enumerate(Y):
            out[i, j] = orc.mash_shared(x, y)
    return out


def _oracle_dist(X, Y):
    out = np.zeros((len(X), len(Y)), np.float64)
    for i, x in enumerate(X):
        for j, y in enumerate(Y):
            out[i, j] = orc.lib().orc_mash_distance(x.ctypes.data, len(x), y.ctypes.data, len(y))
    return out


def test_TestMash(mash):
    """search/mash/mash_test.go:9-62"""
    f1 = mash.New(17, 10)
    f1.Sketch(S62)
    f2 = mash.New(17, 9)
    f2.Sketch(S62)
    assert f1.Distance(f2) == 0
    assert f2.Distance(f1) == 0
    spoofed = mash.New(17, 10)
    spoofed.Sketches[0] = 0  # already zero-filled: mash_test.go:26-39
    assert spoofed.Distance(f1) == 1
    assert f1.Distance(spoofed) == 1
    f3 = mash.New(17, 10)
    f3.Sketch("ATGCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGA")
    f4 = mash.New(17, 5)
    f4.Sketch("ATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGAT")
    # against the oracle on the same sketches (the reference accepts (0.19, 0.21))
    for a, b in ((f3, f4), (f4, f3), (f1, f4), (f4, f2)):
        want = orc.lib().orc_mash_distance(a.Sketches.ctypes.data, a.SketchSize, b.Sketches.ctypes.data, b.SketchSize)
        assert a.Distance(b) == want
        assert a.Similarity(b) == 1 - want or abs(a.Similarity(b) - (1 - want)) < 1e-15


def test_example(mash):
    """search/mash/example_test.go:9-22 prints 0"""
    a = mash.New(17, 10)
    a.Sketch("ATGCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGA")
    b = mash.New(17, 9)
    b.Sketch("ATGCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGA")
    assert a.Distance(b) == 0


def _families(rng, nfam, copies, L, sub, k, s):
    """SURVEY 8d C3 generator in miniature: families of mutated copies, sketched by the oracle."""
    seqs = []
    for _ in range(nfam):
        g = rng.choice(list(b"ACGT"), L).astype(np.uint8)
        for _ in range(copies):
            m = g.copy()
            hit = rng.random(L) < sub
            m[hit] = rng.choice(list(b"ACGT"), int(hit.sum())).astype(np.uint8)
            seqs.append(m.tobytes())
    buf = np.frombuffer(b"".join(seqs), np.uint8)
    offs = np.arange(0, (len(seqs) + 1) * L, L, dtype=np.uint64)
    return orc.mash_sketch_batch(buf, offs, k, s)


def test_allvsall_families_join_path(mash):
    rng = np.random.default_rng(3)
    S = _families(rng, 6, 8, 1500, 0.01, 21, 200)
    counts, dist = mash.distance_matrix_packed(S, S)
    want = _oracle_counts(S, S)
    assert (counts == want).all()
    assert (counts.diagonal() == 200).all() and counts.max() == 200 and (counts == 0).any()
    assert (dist.view(np.uint64) == _oracle_dist(S, S).view(np.uint64)).all()


def test_duplicates_and_different_sizes(mash):
    """multiset semantics of the merge: a value a times in X and b times in Y counts min(a,b)"""
    rng = np.random.default_rng(5)
    X = np.sort(rng.integers(0, 40, (30, 64), dtype=np.uint32), axis=1)   # heavy duplication
    Y = np.sort(rng.integers(0, 40, (25, 48), dtype=np.uint32), axis=1)
    counts, dist = mash.distance_matrix_packed(X, Y)
    assert (counts == _oracle_counts(X, Y)).all()
    assert (dist.view(np.uint64) == _oracle_dist(X, Y).view(np.uint64)).all()
    # hundreds of equal values in one sketch
    X2 = np.sort(rng.integers(0, 3, (4, 900), dtype=np.uint32), axis=1)
    counts2, _ = mash.distance_matrix_packed(X2, X2, True, False)
    assert (counts2 == _oracle_counts(X2, X2)).all()


def test_irregular_sketches_use_reference_loop(mash):
    """positional / unsorted / zero-padded sketches (mash.go:81-84) mixed with regular ones"""
    import torch
    rng = np.random.default_rng(9)
    S = np.sort(rng.integers(0, 1 << 30, (40, 100), dtype=np.uint32), axis=1)
    S[5, :50] = S[6, :50]                                        # two related sketches
    S[5].sort(); S[6].sort()
    S[3] = rng.integers(0, 1 << 30, 100, dtype=np.uint32)       # unsorted
    S[17, 60:] = 0                                               # short sequence: zero tail
    S[29] = 0                                                    # mash.New, never sketched
    want = _oracle_counts(S, S)
    counts, _ = mash.distance_matrix_packed(S, S, True, False)
    assert (counts == want).all()
    dev = torch.device("cuda:0")
    St = torch.from_numpy(S.view(np.int32)).to(dev)
    ct = torch.full((40, 48), -1, dtype=torch.int16, device=dev)[:, :40]  # ld 48 > ny
    work = torch.empty(mash.shared_counts_workspace_bytes(40, 100, 40, 100), dtype=torch.uint8, device=dev)
    mash.shared_counts_dev(St, St, ct, work)
    torch.cuda.synchronize()
    assert (ct.cpu().numpy().view(np.uint16) == want).all()
    mode, ix, iy, ovf, est = mash.shared_counts_mode(work)
    assert (mode, ix, iy, ovf) == (0, 2, 2, 0) and est > 0  # row 29 (all zero) is ascending


def test_dense_input_falls_back_to_merge(mash):
    """many identical sketches: the join estimate exceeds the merge cost -> generic mode, same answer"""
    import torch
    S = np.full((600, 64), 0x096698DE, dtype=np.uint32)  # like the TestMash sketches: one repeated hash
    S[::7, 60:] = 0x0A000000
    S[::5, 0] = 5
    dev = torch.device("cuda:0")
    St = torch.from_numpy(S.view(np.int32)).to(dev)
    ct = torch.zeros((600, 600), dtype=torch.int16, device=dev)
    work = torch.empty(mash.shared_counts_workspace_bytes(600, 64, 600, 64), dtype=torch.uint8, device=dev)
    mash.shared_counts_dev(St, St, ct, work)
    torch.cuda.synchronize()
    assert mash.shared_counts_mode(work)[0] == 1
    assert (ct.cpu().numpy().view(np.uint16) == _oracle_counts(S, S)).all()


def test_row_block_equals_whole(mash):
    """the multi-GPU partition: row blocks against all columns stack to the full matrix"""
    rng = np.random.default_rng(11)
    S = _families(rng, 5, 6, 1200, 0.02, 21, 150)
    whole, _ = mash.distance_matrix_packed(S, S, True, False)
    from poly_amd.sharding import shard_range
    for world in (2, 3):
        rows = []
        for r in range(world):
            lo, hi = shard_range(len(S), r, world)
            rows.append(mash.distance_matrix_packed(S[lo:hi], S, True, False)[0])
        assert (np.vstack(rows) == whole).all()


def test_errors(mash):
    from poly_amd import _lib
    with pytest.raises(_lib.GoPanic):
        mash.New(17, 0).Distance(mash.New(17, 5))


def test_rows_with_many_relatives_overflow_to_merge(mash, join_kind):
    """a row related to more sketches than the join's LDS hash table holds goes to the dense join (16-bit
    counters per column in LDS) -- same counts as the reference's merge"""
    import torch
    rng = np.random.default_rng(13)
    base = np.sort(rng.choice(1 << 31, 64, replace=False).astype(np.uint32))
    S = np.sort(rng.integers(0, 1 << 31, (2400, 64), dtype=np.uint32), axis=1)
    S[:2000, :8] = base[:8]          # 2000 sketches share 8 hashes
    S = np.sort(S, axis=1)
    dev = torch.device("cuda:0")
    St = torch.from_numpy(S.view(np.int32)).to(dev)
    ct = torch.zeros((2400, 2400), dtype=torch.int16, device=dev)
    work = torch.empty(mash.shared_counts_workspace_bytes(2400, 64, 2400, 64), dtype=torch.uint8, device=dev)
    mash.shared_counts_dev(St, St, ct, work)
    torch.cuda.synchronize()
    mode, ix, iy, ovf, est = mash.shared_counts_mode(work)
    assert mode == 0 and (ovf >= 2000 if join_kind == "sparse" else ovf == 0)
    got = ct.cpu().numpy().view(np.uint16)
    rows = list(range(0, 2400, 97)) + [1999, 2000, 2399]
    for i in rows:
        assert (got[i] == _oracle_counts(S[i:i + 1], S)[0]).all()


def test_column_stripes_beyond_one_index(mash, monkeypatch):
    """Y sets of 2^32 hashes and more are joined in column stripes (one index each, side by side in the matrix); the bound
    is lowered here so that the striping runs on a small input: ragged last stripe, ld > ny, irregular sketches, the
    host-pointer flavour with distances"""
    import torch
    rng = np.random.default_rng(21)
    S = _families(rng, 9, 7, 1200, 0.02, 21, 150)           # 63 sketches of 150
    S[10] = rng.integers(0, 1 << 30, 150, dtype=np.uint32)  # unsorted: the merge loop's pairs, in whichever stripe
    X = np.ascontiguousarray(S[5:40])
    want = _oracle_counts(X, S)
    monkeypatch.setenv("POLYHIP_K2_MAX_ITEMS", str(150 * 16 + 7))   # 16 sketches per stripe: 4 stripes, the last of 15
    dev = torch.device("cuda:0")
    Xt, Yt = (torch.from_numpy(a.view(np.int32)).to(dev) for a in (X, S))
    ct = torch.full((35, 80), -1, dtype=torch.int16, device=dev)
    work = torch.empty(mash.shared_counts_workspace_bytes(35, 150, 63, 150), dtype=torch.uint8, device=dev)
    mash.shared_counts_dev(Xt, Yt, ct[:, :63], work)
    torch.cuda.synchronize()
    got = ct.cpu().numpy().view(np.uint16)
    assert (got[:, :63] == want).all() and (got[:, 63:] == 0xFFFF).all()
    counts, dist = mash.distance_matrix_packed(X, S)
    assert (counts == want).all()
    assert (dist.view(np.uint64) == _oracle_dist(X, S).view(np.uint64)).all()
    monkeypatch.delenv("POLYHIP_K2_MAX_ITEMS")
    small = mash.shared_counts_workspace_bytes(35, 150, 16, 150)
    assert work.numel() == small    # the workspace is sized for a stripe, not for the whole set


def test_reuse_after_a_fused_call_with_another_sketch_size(mash, join_kind):
    """round-3 advice: polyhip_mash_shared_counts_dev(X of 500 hashes, Y of 1100) used to leave compact items made for
    10-bit counters (min(sx, sy) <= 1023) that a later polyhip_mash_shared_counts_reuse_dev with sx = 1100 decoded as
    16-bit ones, and with Y sets of two stripes the reuse launched no matching kernel at all.  Reuse after a fused call
    is documented usage (include/polyhip.h): every order of the three calls must give the reference's merge counts."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(35)
    Y = _small_valued_families(rng, 6, 6, 1100)
    X5 = np.sort(np.concatenate([Y[:12, :350], rng.integers(0, 1 << 19, (12, 150), dtype=np.uint32)], axis=1), axis=1)
    Yt = torch.from_numpy(Y.view(np.int32)).to(dev)
    X5t = torch.from_numpy(X5.view(np.int32).copy()).to(dev)
    want5, wantY = _oracle_counts(X5, Y), _oracle_counts(Y, Y)
    for order in (("fused5", "reuse5", "reuseY"), ("fused5", "reuseY", "reuse5"), ("fusedY", "reuse5", "reuseY"),
                  ("build", "reuse5", "reuseY", "reuse5")):
        work = torch.zeros(mash.shared_counts_workspace_bytes(len(Y), 1100, len(Y), 1100), dtype=torch.uint8, device=dev)
        for step in order:
            c5 = torch.full((len(X5), len(Y)), -1, dtype=torch.int16, device=dev)
            cY = torch.full((len(Y), len(Y)), -1, dtype=torch.int16, device=dev)
            if step == "build":
                mash.index_build_dev(Yt, work)
            elif step == "fused5":
                mash.shared_counts_dev(X5t, Yt, c5, work)
            elif step == "fusedY":
                mash.shared_counts_dev(Yt, Yt, cY, work)
            elif step == "reuse5":
                mash.shared_counts_reuse_dev(X5t, Yt, c5, work)
            else:
                mash.shared_counts_reuse_dev(Yt, Yt, cY, work)
            torch.cuda.synchronize()
            if step.endswith("5"):
                assert (c5.cpu().numpy().view(np.uint16) == want5).all(), (order, step)
            elif step != "build":
                assert (cY.cpu().numpy().view(np.uint16) == wantY).all(), (order, step)


def _small_valued_families(rng, nfam, copies, s, bits=19, sub=0.1):
    """related ascending sketches whose hashes stay below 2^bits: a few thousand of them already have fine buckets (the
    value's bits below its bucket are few), which is what lets a SMALL test reach the compact item format"""
    out = []
    for _ in range(nfam):
        base = rng.choice(1 << bits, s, replace=False).astype(np.uint32)
        for _ in range(copies):
            m = base.copy()
            hit = rng.random(s) < sub
            m[hit] = rng.integers(0, 1 << bits, int(hit.sum()), dtype=np.uint32)
            m.sort()
            out.append(m)
    return np.stack(out)


def test_compact_items_where_they_fit_and_a_rebuild_where_not(mash, join_kind):
    """The index build picks 4-byte items on the device when the join to come is the one-stripe dense join and the bits
    fit; a hash repeated many times inside one sketch, or a later join with another counter width, gets 8-byte items
    (the latter by a silent rebuild) -- same counts either way"""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(33)
    S = _small_valued_families(rng, 12, 10, 300)
    St = torch.from_numpy(S.view(np.int32)).to(dev)
    N = len(S)
    work = torch.zeros(mash.shared_counts_workspace_bytes(N, 1200, N, 300), dtype=torch.uint8, device=dev)
    mash.index_build_dev(St, work)
    torch.cuda.synchronize()
    assert mash.index_item_bytes(work) == (4 if join_kind in ("dense", "staged", "twolevel", "zahead") else 8)
    ct = torch.zeros((N, N), dtype=torch.int16, device=dev)
    mash.shared_counts_reuse_dev(St, St, ct, work)
    torch.cuda.synchronize()
    assert (ct.cpu().numpy().view(np.uint16) == _oracle_counts(S, S)).all()
    # X of SketchSize 1200 against the same index: the counts cannot pass 300, same 10-bit counters -> the index stays
    X = np.sort(np.concatenate([S[:20], rng.integers(0, 1 << 19, (20, 900), dtype=np.uint32)], axis=1), axis=1)
    Xt = torch.from_numpy(X.view(np.int32)).to(dev)
    cx = torch.zeros((20, N), dtype=torch.int16, device=dev)
    mash.shared_counts_reuse_dev(Xt, St, cx, work)
    torch.cuda.synchronize()
    assert mash.index_item_bytes(work) == (4 if join_kind in ("dense", "staged", "twolevel", "zahead") else 8)
    assert (cx.cpu().numpy().view(np.uint16) == _oracle_counts(X, S)).all()
    # an index of 1100-hash sketches assumes 16-bit counters; X sketches of 300 hashes need 10-bit ones: not what the
    # build assumed -> rebuilt with 8-byte items, still right
    Y = _small_valued_families(rng, 6, 5, 1100)
    Yt = torch.from_numpy(Y.view(np.int32)).to(dev)
    wY = torch.zeros(mash.shared_counts_workspace_bytes(N, 300, len(Y), 1100), dtype=torch.uint8, device=dev)
    mash.index_build_dev(Yt, wY)
    torch.cuda.synchronize()
    assert mash.index_item_bytes(wY) == (4 if join_kind in ("dense", "staged", "twolevel", "zahead") else 8)
    cyy = torch.zeros((len(Y), len(Y)), dtype=torch.int16, device=dev)
    mash.shared_counts_reuse_dev(Yt, Yt, cyy, wY)          # 16-bit counters, compact items
    torch.cuda.synchronize()
    assert (cyy.cpu().numpy().view(np.uint16) == _oracle_counts(Y, Y)).all()
    cy = torch.zeros((N, len(Y)), dtype=torch.int16, device=dev)
    mash.shared_counts_reuse_dev(St, Yt, cy, wY)
    torch.cuda.synchronize()
    assert mash.index_item_bytes(wY) == 8
    assert (cy.cpu().numpy().view(np.uint16) == _oracle_counts(S, Y)).all()
    # a sketch that repeats one hash 200 times: occurrence numbers beyond what a compact item holds -> 8-byte items
    S2 = S.copy()
    S2[7, :200] = S2[7, 0]
    S2[8, :150] = S2[7, 0]
    S2t = torch.from_numpy(S2.view(np.int32)).to(dev)
    mash.shared_counts_dev(S2t, S2t, ct, work)
    torch.cuda.synchronize()
    assert mash.index_item_bytes(work) == 8
    assert (ct.cpu().numpy().view(np.uint16) == _oracle_counts(S2, S2)).all()
    # a few copies (what the 11 - shift bits number) stay compact
    S3 = S.copy()
    S3[7, :3] = S3[7, 0]
    S3[8, :2] = S3[7, 0]
    S3.sort(axis=1)
    S3t = torch.from_numpy(S3.view(np.int32)).to(dev)
    mash.shared_counts_dev(S3t, S3t, ct, work)
    torch.cuda.synchronize()
    assert mash.index_item_bytes(work) == (4 if join_kind in ("dense", "staged", "twolevel", "zahead") else 8)
    assert (ct.cpu().numpy().view(np.uint16) == _oracle_counts(S3, S3)).all()


def test_c_abi_allgather_one_rank(mash):
    """R1 through the C ABI (polyhip_comm_*: RCCL resolved at run time) on a 1-rank communicator"""
    import ctypes as C
    import torch
    from poly_amd import _lib
    L = _lib.lib()
    ident = (C.c_uint8 * 128)()
    _lib.check(L.polyhip_comm_unique_id(ident))
    comm = C.c_void_p()
    _lib.check(L.polyhip_comm_init_rank(ident, 0, 1, C.byref(comm)))
    assert (L.polyhip_comm_rank(comm), L.polyhip_comm_size(comm)) == (0, 1)
    dev = torch.device("cuda:0")
    local = torch.randint(0, 1 << 31, (300, 64), dtype=torch.int32, device=dev)
    out = torch.zeros_like(local)
    _lib.check(L.polyhip_allgather_sketches_dev(comm, local.data_ptr(), 300, 64, out.data_ptr(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(out, local)
    _lib.check(L.polyhip_comm_destroy(comm))


def test_full_size_config3_row_block_properties(mash, monkeypatch):
    """BASELINE configs[2] at FULL size for one rank of 8: a 12,500 x 100,000 row block of the all-vs-all over
    100,000 sketches of s = 1000 (1000 families x 100 copies at 1 % substitution, sketched by K1).  Properties of
    the whole block: the diagonal shares all s hashes; the block is symmetric where it overlaps its own rows;
    counts never exceed s; 2,400 sampled cells over ALL 100,000 columns (in-family, in-block, and >= 500 in the far
    columns) and 512 full rows equal the reference's merge (mash.go:107-135); distances are exactly 1 - count/s in fp64.
    The WHOLE block is also computed on the two-level index build of round 4 (POLYHIP_K2_B4=0) and must equal the default
    build's (the sliced 4-byte build of round 5) cell for cell."""
    import torch
    from poly_amd import bench_extra
    dev = torch.device("cuda:0")
    s = 1000
    sk = bench_extra.family_sketches(dev, 1000, 100, 10_000, 21, s, seed=0xC3)
    N, nrows = sk.shape[0], sk.shape[0] // 8
    X = sk[:nrows]
    counts = torch.empty((nrows, N), dtype=torch.int16, device=dev)
    work = torch.empty(mash.shared_counts_workspace_bytes(nrows, s, N, s), dtype=torch.uint8, device=dev)
    monkeypatch.delenv("POLYHIP_K2_B4", raising=False)
    mash.shared_counts_dev(X, sk, counts, work)
    torch.cuda.synchronize()
    # the other index build, whole block: 1.25e9 cells, not a sample
    monkeypatch.setenv("POLYHIP_K2_B4", "0")
    counts_two_level = torch.full_like(counts, -1)
    mash.shared_counts_dev(X, sk, counts_two_level, work)
    torch.cuda.synchronize()
    monkeypatch.delenv("POLYHIP_K2_B4", raising=False)
    assert torch.equal(counts, counts_two_level)
    del counts_two_level
    c = counts.to(torch.int32)
    assert bool((c.diagonal() == s).all())
    assert bool((c >= 0).all()) and bool((c <= s).all())
    assert torch.equal(c[:, :nrows], c[:, :nrows].T)
    rng = np.random.default_rng(9)
    sk_h = sk.cpu().numpy().view(np.uint32)  # ALL 100,000 sketches: the sampled columns span the whole block
    c_h = c.cpu().numpy()
    cells = []
    for _ in range(2400):
        i = int(rng.integers(0, nrows))
        u = rng.random()
        if u < 0.35:
            j = (i // 100) * 100 + int(rng.integers(0, 100))  # in the row's own family: hundreds of shared hashes
        elif u < 0.55:
            j = int(rng.integers(0, nrows))                   # the part of the block that overlaps its own rows
        else:
            j = int(rng.integers(nrows, N))                   # far columns: other ranks' sketches
        cells.append((i, j))
    cells += [(0, N - 1), (nrows - 1, N - 1), (nrows - 1, nrows), (0, nrows)]
    assert sum(1 for _, j in cells if j >= nrows) >= 500
    for i, j in cells:
        assert int(c_h[i, j]) == orc.mash_shared(sk_h[i], sk_h[j]), (i, j)
    # 512 FULL rows against the reference's merge over all 100,000 columns (the oracle's C loop on every host core:
    # ctypes releases the GIL): every count, and with it the rows' number of nonzero cells -- a stray count anywhere
    # in a far column cannot hide
    import concurrent.futures as cf
    import os
    rows64 = np.sort(rng.choice(nrows, 512, replace=False))
    groups = np.array_split(rows64, max(1, min(len(rows64), os.cpu_count() or 1)))
    with cf.ThreadPoolExecutor(len(groups)) as ex:
        parts = list(ex.map(lambda g: orc.mash_distance_matrix(sk_h[g], sk_h), groups))
    want_d = np.concatenate(parts, axis=0)
    want_c = np.rint((1.0 - want_d) * s).astype(np.int64)       # exact: Distance = 1 - same/s with same an integer <= s
    assert (1.0 - want_c / np.float64(s) == want_d).all()        # ... and the rounding did not invent a count
    assert (c_h[rows64].astype(np.int64) == want_c).all()
    assert int((c_h[rows64] != 0).sum()) == int((want_c != 0).sum())
    dist = torch.empty((nrows, N), dtype=torch.float64, device=dev)
    mash.distance_from_counts_dev(counts, s, s, dist)
    # exactly mash.go:134,139 -- 1 - float64(same)/float64(size), IEEE division (numpy; torch's scalar divide on the
    # GPU multiplies by the reciprocal, which differs in the last bit)
    rows = rng.choice(nrows, 64, replace=False)
    want = 1.0 - c[torch.from_numpy(rows).to(dev)].cpu().numpy().astype(np.float64) / np.float64(s)
    assert (dist[torch.from_numpy(rows).to(dev)].cpu().numpy() == want).all()
    assert bool((dist[c == 0] == 1.0).all()) and bool((dist.diagonal() == 0.0).all())


def test_dense_rows_over_several_column_stripes(mash, join_kind):
    """70,000 columns need two stripes of the dense join's LDS counters (65,536 columns each); rows with thousands
    of relatives on both sides of the stripe border, duplicates inside sketches (multiset semantics), a few
    irregular (unsorted) Y sketches in between (those pairs take the reference's own loop)."""
    import torch
    rng = np.random.default_rng(17)
    ny, s = 70_000, 16
    Y = np.sort(rng.integers(0, 1 << 31, (ny, s), dtype=np.uint32), axis=1)
    base = np.sort(rng.choice(1 << 31, 6, replace=False).astype(np.uint32))
    rel = rng.choice(ny, 9000, replace=False)            # relatives spread over both stripes
    Y[rel, :4] = base[:4]
    Y[rel[:3000], 4] = base[3]                           # a duplicated value in 3000 of them
    Y = np.sort(Y, axis=1)
    Y[[5, 40_000, 69_999]] = Y[[5, 40_000, 69_999]][:, ::-1]  # irregular: descending
    X = Y[np.concatenate([rel[:40], rng.choice(ny, 24, replace=False)])].copy()
    X[3, :5] = base[3]                                   # X row with a 5-fold duplicate
    X = np.sort(X, axis=1)
    dev = torch.device("cuda:0")
    Xt, Yt = (torch.from_numpy(a.view(np.int32).copy()).to(dev) for a in (X, Y))
    ct = torch.full((len(X), ny), -1, dtype=torch.int16, device=dev)
    work = torch.empty(mash.shared_counts_workspace_bytes(len(X), s, ny, s), dtype=torch.uint8, device=dev)
    mash.shared_counts_dev(Xt, Yt, ct, work)
    torch.cuda.synchronize()
    mode, ix, iy, ovf, est = mash.shared_counts_mode(work)
    assert mode == 0 and iy == 3 and (ovf >= 40 if join_kind == "sparse" else ovf == 0)
    got = ct.cpu().numpy().view(np.uint16)
    assert (got == _oracle_counts(X, Y)).all()


def test_sketches_made_of_repeated_hashes_stay_in_the_join(mash):
    """homopolymer / tandem-repeat reads sketch to a few hashes repeated hundreds of times (the reference keeps
    duplicates).  Such sketches are regular for the join (the occurrence number has 32 - ceil(log2(ny)) bits), and
    min(multiplicity in X, multiplicity in Y) per shared value is what the reference's merge counts."""
    import torch
    rng = np.random.default_rng(23)
    s, ny = 1000, 300
    Y = np.sort(rng.integers(0, 1 << 30, (ny, s), dtype=np.uint32), axis=1)
    h1, h2, h3 = 12345, 777777, 900000001
    Y[0, :] = h1                                  # 1000 copies of one hash
    Y[1, :600] = h1; Y[1, 600:] = h2              # 600 + 400
    Y[2, :300] = h1; Y[2, 300:650] = h2; Y[2, 650:] = h3
    Y[3, :257] = h2                               # just above the old 256 limit
    Y[4, :5] = h3
    Y = np.sort(Y, axis=1)
    X = Y[[0, 1, 2, 3, 4, 50, 51]].copy()
    dev = torch.device("cuda:0")
    Xt, Yt = (torch.from_numpy(a.view(np.int32).copy()).to(dev) for a in (X, Y))
    ct = torch.full((len(X), ny), -1, dtype=torch.int16, device=dev)
    work = torch.empty(mash.shared_counts_workspace_bytes(len(X), s, ny, s), dtype=torch.uint8, device=dev)
    mash.shared_counts_dev(Xt, Yt, ct, work)
    torch.cuda.synchronize()
    mode, ix, iy, ovf, est = mash.shared_counts_mode(work)
    assert (mode, ix, iy) == (0, 0, 0)
    got = ct.cpu().numpy().view(np.uint16)
    assert (got == _oracle_counts(X, Y)).all()
    assert got[0, 0] == 1000 and got[0, 1] == 600 and got[1, 2] == 300 + 350


def test_index_is_built_once_and_reused_for_row_blocks(mash, join_kind):
    """polyhip_mash_index_build_dev + polyhip_mash_shared_counts_reuse_dev: Y's inverted index is built once, then
    row blocks of different sizes (and an unrelated X with irregular rows) are joined against it -- every block equals
    the one-shot call and the oracle; also SketchSize 1500 (16-bit dense counters instead of 10-bit)"""
    import torch
    rng = np.random.default_rng(31)
    dev = torch.device("cuda:0")
    for s, ny in ((200, 3000), (1500, 700)):
        fam = np.sort(rng.integers(0, 1 << 30, (ny // 20, s), dtype=np.uint32), axis=1)
        Y = np.repeat(fam, 20, axis=0)
        mut = rng.random(Y.shape) < 0.1
        Y[mut] = rng.integers(0, 1 << 30, int(mut.sum()), dtype=np.uint32)
        Y = np.sort(Y, axis=1)
        Y[7] = Y[7][::-1]                      # one irregular column
        Yt = torch.from_numpy(Y.view(np.int32).copy()).to(dev)
        work = torch.empty(mash.shared_counts_workspace_bytes(ny, s, ny, s), dtype=torch.uint8, device=dev)
        whole = torch.full((ny, ny), -1, dtype=torch.int16, device=dev)
        mash.shared_counts_dev(Yt, Yt, whole, work)
        torch.cuda.synchronize()
        whole = whole.cpu().numpy().view(np.uint16)
        for i in list(range(0, ny, max(1, ny // 9))) + [7]:
            assert (whole[i] == _oracle_counts(Y[i:i + 1], Y)[0]).all(), (s, i)
        mash.index_build_dev(Yt, work)
        for lo, hi in ((0, 1), (1, 600), (600, ny - 3), (ny - 3, ny)):
            blk = torch.full((hi - lo, ny), -1, dtype=torch.int16, device=dev)
            mash.shared_counts_reuse_dev(Yt[lo:hi], Yt, blk, work)
            torch.cuda.synchronize()
            assert (blk.cpu().numpy().view(np.uint16) == whole[lo:hi]).all(), (s, lo, hi)
        X = np.sort(rng.integers(0, 1 << 30, (50, s), dtype=np.uint32), axis=1)
        X[:10] = Y[rng.choice(ny, 10)]
        X[3] = X[3][::-1]
        blk = torch.full((50, ny), -1, dtype=torch.int16, device=dev)
        mash.shared_counts_reuse_dev(torch.from_numpy(X.view(np.int32).copy()).to(dev), Yt, blk, work)
        torch.cuda.synchronize()
        assert (blk.cpu().numpy().view(np.uint16) == _oracle_counts(X, Y)).all()
