"""Larger randomized parity sweeps (seconds each): the same C-ABI entry points against the oracle on
inputs that mix the regular case with the reference's quirks, so that rarely taken device paths
(K1's redo list, K2's overflow rows / irregular sketches, K5's serial fallback) are exercised together."""
import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


def _pack(seqs):
    offs = np.zeros(len(seqs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    return np.frombuffer(b"".join(seqs), np.uint8).copy(), offs


def _mixed_reads(rng, n, maxlen):
    out = []
    for i in range(n):
        L = int(rng.integers(0, maxlen))
        kind = i % 9
        if kind == 0:
            out.append(bytes(rng.choice(list(b"AC"), L).astype(np.uint8)))            # low complexity
        elif kind == 1:
            unit = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 40))).astype(np.uint8))
            out.append((unit * (L // len(unit) + 1))[:L])                             # tandem repeat
        elif kind == 2:
            out.append(bytes(rng.integers(0, 256, L, dtype=np.uint8)))                # arbitrary bytes
        else:
            out.append(bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8)))
    return out


@pytest.mark.parametrize("k,s", [(21, 1000), (17, 64), (31, 300), (13, 2500), (4, 50)])
def test_sketch_mixed_batch(k, s):
    from poly_amd import mash
    rng = np.random.default_rng(k * 1000 + s)
    reads = _mixed_reads(rng, 700, 30_000)
    buf, offs = _pack(reads)
    prior = rng.integers(0, 1 << 32, (len(reads), s), dtype=np.uint32)  # stale Sketches must survive where the reference leaves them
    got = mash.sketch_batch_packed(buf, offs, k, s, out=prior.copy())
    want = prior.copy()
    for i, r in enumerate(reads):
        orc.lib().orc_mash_sketch(np.frombuffer(r, np.uint8).ctypes.data if r else None, len(r), k, s,
                                  want[i].ctypes.data, 0)
    assert (got == want).all()


def test_distance_mixed_sets():
    from poly_amd import mash
    rng = np.random.default_rng(77)
    s = 96
    base = [np.sort(rng.integers(0, 1 << 32, s, dtype=np.uint32)) for _ in range(40)]
    X = []
    for i in range(700):
        b = base[i % 40].copy()
        b[rng.integers(0, s, int(rng.integers(0, 30)))] = rng.integers(0, 1 << 32, 1, dtype=np.uint32)
        b.sort()
        if i % 97 == 0:
            rng.shuffle(b)          # unsorted
        if i % 131 == 0:
            b[s // 2:] = 0          # zero tail
        if i % 53 == 0:
            b[:] = b[0]             # one repeated hash
        X.append(b)
    X = np.stack(X)
    counts, dist = mash.distance_matrix_packed(X[:300], X)
    for i in range(0, 300, 7):
        for j in range(0, 700, 3):
            assert counts[i, j] == orc.mash_shared(np.ascontiguousarray(X[i]), np.ascontiguousarray(X[j])), (i, j)
    assert ((1 - counts / s) == dist).all()


def test_rotation_and_hash_mixed():
    from poly_amd import seqhash
    rng = np.random.default_rng(5)
    seqs = [s.upper().replace(b"\x00", b"A") for s in _mixed_reads(rng, 300, 6000)]
    seqs = [bytes(c if c in b"ACGTUNRYSWKMBDHVZ" else ord("A") for c in s) for s in seqs]
    got = seqhash.HashBatch(seqs, "DNA", True, True)
    for s_, g in zip(seqs, got):
        assert g == orc.seqhash(s_, "DNA", True, True)
    rot = seqhash.RotateBatch([s.decode("latin-1") for s in seqs])
    for s_, r in zip(seqs, rot):
        assert r.encode("latin-1") == orc.rotate_sequence(s_)


@pytest.mark.parametrize("short_reads", [False, True])
def test_sketch_host_flavour_pipelines_chunks(short_reads):
    """polyhip_mash_sketch_batch (host pointers) streams a batch larger than its 256 MB chunk through two
    slots; every row equals the device flavour's, and rows the reference leaves (partly) untouched keep the
    caller's prior Sketches -- with short reads in the batch the prior rows are uploaded, without them not."""
    import torch
    from poly_amd import mash
    n, L, k, s = 72_000, 10_000, 21, 1000  # 720 MB of reads + 288 MB of sketches: 4 chunks
    dev = torch.device("cuda:0")
    seqs_t = torch.empty(n * L, dtype=torch.uint8, device=dev)
    mash.synth_dna_dev(0x51, seqs_t)
    seqs = seqs_t.cpu().numpy()
    lens = np.full(n, L, np.uint64)
    if short_reads:
        lens[[5, 30_000, n - 1]] = [500, 10, 1020]  # positional / untouched / exactly k + s - 1
    # ragged offsets into the same byte stream (reads simply start where the previous one ended)
    offs = np.zeros(n + 1, np.uint64)
    offs[1:] = np.cumsum(lens)
    prior = np.full((n, s), 0xABCD0000, np.uint32) + np.arange(s, dtype=np.uint32)
    got = mash.sketch_batch_packed(seqs, offs, k, s, out=prior.copy())
    want_t = torch.from_numpy(prior.view(np.int32).copy()).to(dev)
    mash.sketch_batch_dev(seqs_t, torch.from_numpy(offs.view(np.int64)).to(dev), k, s, want_t)
    want = want_t.cpu().numpy().view(np.uint32)
    assert (got == want).all()
    for r in (0, 5, 30_000, n - 1):  # and against the oracle on the special rows
        one = np.ascontiguousarray(seqs[int(offs[r]):int(offs[r + 1])])
        ref = orc.mash_sketch_batch(one, np.array([0, len(one)], np.uint64), k, s, out=prior[r:r + 1].copy())
        assert (got[r] == ref[0]).all(), r
