"""Randomised K1/K2 check against the oracle (not part of the test suite; run on the GPU box):
random k, s, read lengths and read kinds; every sketch must equal the oracle's, shared counts too."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import oracle as orc
from poly_amd import mash
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed0 + it)
    it += 1
    k = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 11, 16, 17, 20, 21, 22, 23, 31, 32, 33, 47, 64]))
    s = int(rng.choice([2, 3, 10, 64, 200, 1000, 1001, 1500, 2048, 4000]))
    n = int(rng.integers(1, 60))
    maxlen = int(rng.choice([50, 400, 3000, 12000, 40000]))
    reads = []
    for i in range(n):
        L = int(rng.integers(0, maxlen))
        kind = int(rng.integers(0, 6))
        if kind == 0:
            r = bytes(rng.choice(list(b"AC"), L).astype(np.uint8))
        elif kind == 1:
            unit = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 30))).astype(np.uint8))
            r = (unit * (L // len(unit) + 1))[:L]
        elif kind == 2:
            r = bytes(rng.integers(0, 256, L, dtype=np.uint8))
        else:
            r = bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8))
        reads.append(r)
    offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum([len(r) for r in reads])
    buf = np.frombuffer(b"".join(reads) + b"\0", np.uint8)[:-1].copy()
    prior = rng.integers(0, 2**32, (n, s), dtype=np.uint32)
    want = orc.mash_sketch_batch(buf, offs, k, s, out=prior.copy())
    got = mash.sketch_batch_packed(buf, offs, k, s, out=prior.copy())
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) == 0, ("sketch", it, k, s, [(int(b), len(reads[b])) for b in bad[:5]])
    if n >= 2:
        # Distance on whatever Sketch left (positional rows are unsorted: the reference's merge runs on them as is)
        cnt, _ = mash.distance_matrix_packed(got, got, want_counts=True, want_dist=False)
        for _ in range(30):
            i, j = int(rng.integers(0, n)), int(rng.integers(0, n))
            assert int(cnt[i, j]) == orc.mash_shared(got[i], got[j]), ("shared", it, k, s, i, j)
    if it % 20 == 0:
        print(f"it {it}: k {k} s {s} n {n} maxlen {maxlen} ok", flush=True)
print("fuzz done", it, "iterations")
