"""Randomised sweep of K5 (least rotation + rotated copy) against the oracle's Booth: lengths 0 .. beyond what LDS holds,
alphabets of 1..256 symbols, runs of the least byte, tandem repeats that do / do not close, near-periodic input, every
alignment of a sequence inside the packed buffer.  usage: fuzz_k5.py [seconds] [seed]  (GPU box, not part of the suite)"""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
import oracle as orc
from poly_amd import seqhash
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
it = nseq = 0


def one(rng):
    kind = int(rng.integers(0, 9))
    L = int(rng.choice([0, 1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 23, 24, 25, 31, 32, 33, 47, 48, 63, 64, 65, 100, 255, 256, 257, 1000, 1023, 1024,
                        1025, 4999, 5000, 20000, int(rng.integers(0, 3000))]))
    if rng.random() < 0.02:
        L = int(rng.choice([122_000, 122_860, 122_870, 122_880, 123_000, 140_000, 70_000]))
    if kind == 0:
        alpha = list(b"ACGT")
    elif kind == 1:
        alpha = list(b"AC")
    elif kind == 2:
        alpha = list(range(256))
    elif kind == 3:
        alpha = list(b"A")
    else:
        alpha = list(b"ACGT")
    q = bytearray(rng.choice(alpha, L).astype(np.uint8).tobytes())
    if kind == 4 and L:                     # runs of the least byte
        for _ in range(int(rng.integers(1, 6))):
            a = int(rng.integers(0, L))
            b = min(L, a + int(rng.integers(1, max(2, L // 2))))
            q[a:b] = b"A" * (b - a)
    elif kind == 5 and L:                   # tandem repeat, closes
        u = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, max(2, min(L, 200))))).astype(np.uint8))
        q = bytearray(u * max(1, L // len(u)))
    elif kind == 6 and L:                   # tandem repeat, does not close
        u = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, max(2, min(L, 200))))).astype(np.uint8))
        q = bytearray((u * (L // len(u) + 1))[:L])
    elif kind == 7 and L:                   # near-periodic: one byte changed
        u = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, max(2, min(L, 60))))).astype(np.uint8))
        q = bytearray(u * max(1, L // len(u)))
        q[int(rng.integers(0, len(q)))] = int(rng.choice(list(b"ACGTN")))
    elif kind == 8 and L > 8:               # two copies of a random half (period n / 2), sometimes broken at the end
        h = bytes(q[: L // 2])
        q = bytearray(h + h)
        if rng.random() < 0.5:
            q[-1] = q[-1] ^ 1
    return bytes(q)


while time.time() < t_end:
    rng = np.random.default_rng(seed0 * 1_000_003 + it)
    it += 1
    seqs = [one(rng) for _ in range(int(rng.integers(1, 40)))]
    offs = np.zeros(len(seqs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(q) for q in seqs])
    buf = np.frombuffer(b"".join(seqs) + b"\0", np.uint8)[:-1].copy()
    want_rot = rng.random() < 0.8
    rot, out = seqhash.least_rotation_batch_packed(buf, offs, want_rot)
    for i, q in enumerate(seqs):
        assert int(rot[i]) == orc.booth_least_rotation(q), ("k5 index", it, i, len(q), int(rot[i]))
        if want_rot:
            assert out[int(offs[i]):int(offs[i + 1])].tobytes() == orc.rotate_sequence(q), ("k5 rotated", it, i, len(q))
    nseq += len(seqs)
print("k5 fuzz done:", it, "batches,", nseq, "sequences, no mismatch")
