"""K1 parity: poly_amd.mash (HIP, through the C ABI) vs the CPU oracle, bit-exact.

Mirrors search/mash/mash_test.go where the reference has a test, then widens
to ragged / empty / degenerate batches and size-independent properties."""
import hashlib
import os

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SEQ1 = "ATGCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGA"
SEQ2 = "ATCGATCGATCGATCGATCGATCGATCGATCGATCGAATGCGATCGATCGATCGATCGATCG"


@pytest.fixture(scope="module")
def mash():
    from poly_amd import mash as m
    return m


def _pack(seqs):
    offs = np.zeros(len(seqs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    return np.frombuffer(b"".join(seqs), np.uint8).copy(), offs


def _check_batch(mash, seqs, k, s, prior=None):
    buf, offs = _pack(seqs)
    n = len(seqs)
    init = np.zeros((n, s), np.uint32) if prior is None else prior
    want = orc.mash_sketch_batch(buf, offs, k, s, out=init.copy())
    got = mash.sketch_batch_packed(buf, offs, k, s, out=init.copy())
    bad = np.nonzero((want != got).any(axis=1))[0]
    assert bad.size == 0, f"k={k} s={s}: {bad.size} sketches differ, first {bad[0]} (len {len(seqs[bad[0]])})"


def test_TestMash_sketches(mash):
    """search/mash/mash_test.go:9-62 -- the sketches behind its distance assertions"""
    f1 = mash.New(17, 10)
    f1.Sketch(SEQ1)
    assert list(f1.Sketches) == [0x096698DE] * 10  # duplicates kept
    f2 = mash.New(17, 9)
    f2.Sketch(SEQ1)
    assert list(f2.Sketches) == [0x096698DE] * 9
    f3 = mash.New(17, 5)
    f3.Sketch(SEQ2)
    assert list(f3.Sketches) == [0x08F7DC27] + [0x096698DE] * 4


def test_phix174_config1(mash):
    """BASELINE config 1 input (data/phix174.gb, k=21, s=1000) on the GPU path."""
    seq = open(os.path.join(GOLD, "phix174.seq")).read().strip()
    m = mash.New(21, 1000)
    m.Sketch(seq)
    assert hashlib.sha256(m.Sketches.astype("<u4").tobytes()).hexdigest() == \
        "943c9bb7559e8cb151b9382dbdab7ff8d1b642f64ad1ec7b9b03d709f8ad898a"
    o = orc.Mash(21, 1000)
    o.Sketch(seq)
    assert (o.Sketches == m.Sketches).all()


@pytest.mark.parametrize("k,s", [(21, 1000), (17, 10), (31, 100), (4, 16), (3, 7), (1, 2), (0, 5),
                                 (16, 64), (23, 257), (64, 500), (22, 1000), (19, 2048)])
def test_random_ragged_batch(mash, k, s):
    rng = np.random.default_rng(k * 1000 + s)
    lens = rng.integers(0, 6000, 200).tolist() + [0, 1, k, k + 1, k + s - 1, k + s, k + s + 1, 10_000, 2048 + k, 2049 + k]
    stream = orc.synth_dna(0xABC + k, int(sum(lens)))
    seqs, p = [], 0
    for L in lens:
        seqs.append(stream[p:p + L].tobytes())
        p += L
    prior = rng.integers(0, 2**32, (len(seqs), s), dtype=np.uint32)  # stale state must survive where Go leaves it
    _check_batch(mash, seqs, k, s, prior=prior)


def test_arbitrary_bytes_and_case(mash):
    """no case folding, no N filtering: raw bytes are hashed (mash.go:74-76)"""
    rng = np.random.default_rng(5)
    seqs = [rng.integers(0, 256, int(L), dtype=np.uint8).tobytes() for L in rng.integers(30, 5000, 64)]
    seqs += [b"acgtn" * 700, b"ACGTN" * 700]
    _check_batch(mash, seqs, 21, 300)


def test_degenerate_low_complexity(mash):
    """all hashes equal / few distinct hashes: the threshold guess fails and the
    exact accept-everything path + duplicate ranking must take over"""
    seqs = [b"A" * 10_000, b"AC" * 5_000, b"ACG" * 3000, b"A" * 4000 + b"C" * 4000, b"ACGTTGCA" * 1500,
            b"A" * 1021, b"A" * 1022, b"T" * 30_000]
    for k, s in [(21, 1000), (17, 10), (21, 2)]:
        _check_batch(mash, seqs, k, s)


def test_long_sequence_many_tiles(mash):
    """genome-scale input: hundreds of tiles, shrinks in both modes"""
    seq = orc.synth_dna(0x600D, 700_000).tobytes()
    _check_batch(mash, [seq, seq[:300_001], seq[5:123_456]], 21, 1000)
    # adversarial for the threshold guess: a uniform half followed by a half whose
    # hashes are all tiny cannot be built cheaply, so force many shrinks instead with small s
    _check_batch(mash, [seq], 21, 2)
    _check_batch(mash, [seq[:200_000]], 31, 8192)


def test_device_resident_matches_host_and_oracle(mash):
    import torch
    dev = torch.device("cuda:0")
    n, L, k, s = 512, 10_000, 21, 1000
    seqs = torch.empty(n * L, dtype=torch.uint8, device=dev)
    mash.synth_dna_dev(0xC2, seqs)
    host = orc.synth_dna(0xC2, n * L)
    assert (seqs.cpu().numpy() == host).all()  # GPU generator == oracle generator
    offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    out = torch.zeros((n, s), dtype=torch.int32, device=dev)
    mash.sketch_batch_dev(seqs, offs, k, s, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32)
    want = orc.mash_sketch_batch(host, offs.cpu().numpy().astype(np.uint64), k, s)
    assert (got == want).all()
    # size-independent properties at the full read size: ascending, and the
    # sketch of a read is the bottom-s of the multiset of its window hashes
    assert (np.diff(got.astype(np.int64), axis=1) >= 0).all()


def test_sketch_is_idempotent_and_order_free(mash):
    """Sketch overwrites the whole slice when n-k >= s: calling it twice, or on a
    zeroed vs dirty slice, gives the same sketch"""
    seq = orc.synth_dna(77, 20_000).tobytes()
    a = mash.New(21, 1000)
    a.Sketch(seq)
    first = a.Sketches.copy()
    a.Sketch(seq)
    assert (a.Sketches == first).all()
    b = mash.New(21, 1000)
    b.Sketches[:] = 0xFFFFFFFF
    b.Sketch(seq)
    assert (b.Sketches == first).all()


def test_sketchsize_below_two_is_the_reference_read_by_read(mash):
    """mash.go:96,98 index Sketches[-1]: with s == 0 a sequence panics iff it has a window; with s == 1 window 0 fills
    Sketches[0] and the sequence panics iff a LATER window hashes below it.  The batch call reports the FIRST such
    sequence and writes the rows of the ones that do not panic exactly as the reference leaves them."""
    from poly_amd import _lib
    k = 21
    rng = np.random.default_rng(5)
    blob = orc.synth_dna(0x51, 200_000).tobytes()

    def ref(seq, s, prior):
        m = orc.Mash(k, s)
        m.Sketches[:] = prior
        try:
            m.Sketch(seq)
        except orc.GoPanic:
            return None
        return m.Sketches.copy()

    # ---- s == 1: reads whose FIRST window holds the smallest hash do not panic
    safe, unsafe = [], []
    for i in range(400):
        L = int(rng.integers(k + 2, 400))
        a = int(rng.integers(0, len(blob) - L))
        q = blob[a:a + L]
        hs = [orc.murmur3_32(q[w:w + k]) for w in range(L - k)]
        (safe if min(hs) == hs[0] else unsafe).append(q)
    short = [b"", b"ACGT", blob[:k], blob[:k + 1]]          # no window (x3), exactly one window
    assert len(safe) >= 3 and len(unsafe) >= 3
    batch = safe + short
    prior = rng.integers(1, 1 << 32, (len(batch), 1), dtype=np.uint32)
    buf, offs = _pack(batch)
    got = mash.sketch_batch_packed(buf, offs, k, 1, out=prior.copy())
    for i, q in enumerate(batch):
        want = ref(q, 1, prior[i])
        assert want is not None and got[i, 0] == want[0], i
    mixed = safe[:2] + [unsafe[0]] + safe[2:] + [unsafe[1]]
    buf, offs = _pack(mixed)
    with pytest.raises(_lib.GoPanic) as ei:
        mash.sketch_batch_packed(buf, offs, k, 1)
    assert "sequence 2 " in str(ei.value) and "mash.go:98" in str(ei.value)   # the first one the reference panics on
    assert ref(mixed[2], 1, 0) is None and ref(mixed[0], 1, 0) is not None
    # ---- s == 0: only sequences with a window panic
    none = [b"", b"A" * k, blob[:5]]
    buf, offs = _pack(none)
    mash.sketch_batch_packed(buf, offs, k, 0, out=np.zeros((3, 0), np.uint32))     # len <= k everywhere: fine
    for q in none:
        assert ref(q, 0, 0) is not None
    buf, offs = _pack(none + [blob[:k + 1]])
    with pytest.raises(_lib.GoPanic) as ei:
        mash.sketch_batch_packed(buf, offs, k, 0, out=np.zeros((4, 0), np.uint32))
    assert "sequence 3 " in str(ei.value) and "mash.go:96" in str(ei.value)
    assert ref(blob[:k + 1], 0, 0) is None
    # the single-call API: the reference's own panics
    with pytest.raises(_lib.GoPanic):
        mash.New(21, 1).Sketch(unsafe[0])
    with pytest.raises(_lib.GoPanic):
        mash.New(21, 0).Sketch("ACGT" * 100)
    one = mash.New(21, 1)
    one.Sketch(safe[0])
    assert one.Sketches[0] == orc.murmur3_32(safe[0][:k])


@pytest.mark.parametrize("k,s", [(21, 8193), (21, 10000), (21, 65535), (17, 20000), (4097, 1000), (10000, 64), (5000, 9000),
                                 (4096, 8192), (31, 16384)])
def test_sketch_and_kmer_sizes_beyond_the_lds_kernels(mash, k, s):
    """mash.New(21, 10000) is ordinary usage and the reference takes any KmerSize: what the LDS-resident kernels cannot
    hold goes through the wide kernel (windows hashed from global memory, candidates in a stream-ordered scratch) --
    same sketches as the oracle, incl. reads with fewer windows than SketchSize, duplicates and dirty prior state."""
    rng = np.random.default_rng(k * 131 + s)
    blob = orc.synth_dna(k + s, 400_000).tobytes()
    L1 = k + s + 5000
    seqs = [blob[:L1 + 120_000], blob[7:7 + L1], blob[3:3 + k + s], blob[11:11 + k + s - 1], blob[:k], blob[:k + 1], b"",
            (b"ACGTTGCA" * ((L1 + 40_000) // 8 + 1))[:L1 + 40_000],            # period 8: at most 8 distinct hashes
            b"A" * (k + s + 300)]                                              # one hash, s + 300 times
    prior = rng.integers(0, 1 << 32, (len(seqs), s), dtype=np.uint32)
    _check_batch(mash, seqs, k, s, prior=prior)


def test_wide_kernel_device_resident(mash):
    """the _dev flavour of the wide path (stream-ordered scratch) on many reads at once"""
    import torch
    dev = torch.device("cuda:0")
    n, L, k, s = 300, 30_000, 21, 10_000
    seqs = torch.empty(n * L, dtype=torch.uint8, device=dev)
    mash.synth_dna_dev(0xC7, seqs)
    offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    out = torch.zeros((n, s), dtype=torch.int32, device=dev)
    mash.sketch_batch_dev(seqs, offs, k, s, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32)
    host = orc.synth_dna(0xC7, n * L)
    rows = [0, 1, 150, 299]
    for r in rows:
        want = orc.mash_sketch_batch(host[r * L:(r + 1) * L], np.array([0, L], np.uint64), k, s)
        assert (got[r] == want[0]).all()
    assert (np.diff(got.astype(np.int64), axis=1) >= 0).all()


def test_full_size_config2_properties(mash):
    """BASELINE configs[1] at FULL size (1,000,000 reads x 10 kb, k=21, s=1000; 14 GB resident): size-independent
    properties of the whole output -- every row strictly usable by Distance (ascending), every row independent
    of its batch (equal to the same read sketched in a batch of 64), 16 sampled rows equal to the oracle's faithful
    (sort-on-accept) variant, ALL 1,000,000 rows equal to the oracle's tight variant, and the second run bit-identical
    (no leftover state)."""
    import torch
    dev = torch.device("cuda:0")
    n, L, k, s = 1_000_000, 10_000, 21, 1000
    seqs = torch.empty(n * L, dtype=torch.uint8, device=dev)
    mash.synth_dna_dev(0xC2, seqs)
    offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    out = torch.zeros((n, s), dtype=torch.int32, device=dev)
    mash.sketch_batch_dev(seqs, offs, k, s, out)
    torch.cuda.synchronize()
    # hashes are unsigned: compare through int64
    step = 100_000
    for r0 in range(0, n, step):
        blk = out[r0:r0 + step].to(torch.int64) & 0xFFFFFFFF
        assert bool((blk[:, 1:] >= blk[:, :-1]).all()), r0
    rng = np.random.default_rng(2)
    pick = np.sort(rng.choice(n, 64, replace=False))
    sub = torch.cat([seqs[int(r) * L:(int(r) + 1) * L] for r in pick])
    sub_out = torch.zeros((64, s), dtype=torch.int32, device=dev)
    mash.sketch_batch_dev(sub, offs[:65].contiguous(), k, s, sub_out)
    assert torch.equal(sub_out, out[torch.from_numpy(pick).to(dev)])
    host = sub[:16 * L].cpu().numpy()
    want = orc.mash_sketch_batch(host, np.arange(0, 17 * L, L, dtype=np.uint64), k, s, faithful=True)
    assert (sub_out[:16].cpu().numpy().view(np.uint32) == want).all()
    # EVERY row against the oracle (round-4 verdict: 16 rows say nothing about read 999,999): the oracle's tight variant --
    # equal to the faithful one by tests/test_oracle_golden.py::test_tight_sketch_variant_equals_... -- on every host core
    # (ctypes releases the GIL), chunk by chunk so that the host holds 0.5 GB of reads at a time
    import concurrent.futures as cf
    import os
    ncpu = max(1, min(os.cpu_count() or 1, 64))
    chunk = 50_000
    loffs = np.arange(0, (chunk + 1) * L, L, dtype=np.uint64)
    with cf.ThreadPoolExecutor(ncpu) as ex:
        for r0 in range(0, n, chunk):
            m = min(chunk, n - r0)
            h = seqs[r0 * L:(r0 + m) * L].cpu().numpy()
            got = out[r0:r0 + m].cpu().numpy().view(np.uint32)
            cuts = [m * t // ncpu for t in range(ncpu + 1)]

            def one(t):
                a, b = cuts[t], cuts[t + 1]
                if a == b:
                    return -1
                w = orc.mash_sketch_batch(h[a * L:b * L], loffs[:b - a + 1], k, s)
                bad = np.nonzero((w != got[a:b]).any(axis=1))[0]
                return r0 + a + int(bad[0]) if len(bad) else -1
            bad = [b for b in ex.map(one, range(ncpu)) if b >= 0]
            assert not bad, f"row {min(bad)} of the full-size batch differs from the oracle"
    again = torch.zeros((n, s), dtype=torch.int32, device=dev)
    mash.sketch_batch_dev(seqs, offs, k, s, again)
    assert torch.equal(out, again)


def test_repeated_kmers_big_bins(mash):
    """reads with a random part and a tandem-repeated part (copy numbers around and above the 32-value bin
    limit): the threshold pass still succeeds, but its counting sort meets bins holding dozens of EQUAL hashes,
    which whole waves place (rank_big_bins); plus pure repeats / homopolymers / poly-A tails that go to the
    general kernel.  Every sketch equals the oracle's (duplicates kept, as the reference keeps them)."""
    rng = np.random.default_rng(33)
    reads = []
    for copies in (20, 31, 32, 33, 40, 64, 100, 300):
        for unit_len in (37, 200, 410):
            unit = bytes(rng.choice(list(b"ACGT"), unit_len).astype(np.uint8))
            rep = unit * copies
            for rnd_len in (0, 1500, 6000):
                rnd = bytes(rng.choice(list(b"ACGT"), rnd_len).astype(np.uint8))
                reads.append((rnd + rep)[:12000])
                reads.append((rep[: len(rep) // 2] + rnd + rep[len(rep) // 2:])[:12000])
    reads += [b"A" * 10000, b"AC" * 5000, bytes(rng.choice(list(b"ACGT"), 5000).astype(np.uint8)) + b"A" * 5000]
    buf, offs = _pack(reads)
    for k, s in ((21, 1000), (17, 200), (31, 2000)):
        got = mash.sketch_batch_packed(buf, offs, k, s)
        want = orc.mash_sketch_batch(buf, offs, k, s)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert len(bad) == 0, (k, s, bad[:8], [len(reads[b]) for b in bad[:8]])
