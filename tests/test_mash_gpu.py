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


def test_error_paths(mash):
    from poly_amd import _lib
    with pytest.raises(_lib.GoPanic):
        mash.New(21, 1).Sketch("ACGT" * 100)
    with pytest.raises(_lib.GoPanic):
        mash.New(21, 0).Sketch("ACGT" * 100)


def test_full_size_config2_properties(mash):
    """BASELINE configs[1] at FULL size (1,000,000 reads x 10 kb, k=21, s=1000; 14 GB resident): size-independent
    properties of the whole output -- every row strictly usable by Distance (ascending), every row independent
    of its batch (equal to the same read sketched in a batch of 64), 16 sampled rows equal to the oracle, and
    the second run bit-identical (no leftover state)."""
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
    want = orc.mash_sketch_batch(host, np.arange(0, 17 * L, L, dtype=np.uint64), k, s)
    assert (sub_out[:16].cpu().numpy().view(np.uint32) == want).all()
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
