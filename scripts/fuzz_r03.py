"""Randomised sweep over what round 3 added (run on the GPU box, not part of the suite), everything against the oracle:
K1's wide kernel (SketchSize up to 65535, KmerSize up to 6000) and SketchSize 0 / 1 read by read; K2's compact items, the
index built in parts, column stripes; the output-driven FASTA gather; K5's one-block search on exact tandem repeats."""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, '.')
import oracle as orc
from oracle import fasta_ref as fr
from poly_amd import _lib, fasta, mash, seqhash
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
dev = torch.device('cuda:0')
it = 0
stats = {"wide": 0, "tiny_s": 0, "compact": 0, "wide_items": 0, "parts": 0, "stripes": 0, "fasta": 0, "k5": 0}


def pack(seqs):
    offs = np.zeros(len(seqs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(q) for q in seqs])
    return np.frombuffer(b"".join(seqs) + b"\0", np.uint8)[:-1].copy(), offs


def some_read(rng, L):
    kind = int(rng.integers(0, 6))
    if kind == 0:
        return bytes(rng.choice(list(b"AC"), L).astype(np.uint8))
    if kind == 1:
        unit = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 30))).astype(np.uint8))
        return (unit * (L // len(unit) + 1))[:L]
    if kind == 2:
        return bytes(rng.integers(0, 256, L, dtype=np.uint8))
    return bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8))


while time.time() < t_end:
    rng = np.random.default_rng(seed0 + it)
    it += 1
    # ---- K1: beyond the LDS kernels, and SketchSize 0 / 1
    if rng.random() < 0.5:
        k = int(rng.choice([1, 5, 17, 21, 31, 64, 4097, 5000, 6000]))
        s = int(rng.choice([8193, 9000, 12000, 20000, 65535])) if k < 100 or rng.random() < 0.3 else int(rng.choice([2, 64, 1000, 9000]))
        if k < 100 and rng.random() < 0.2:
            s = int(rng.choice([64, 1000]))          # (the LDS kernels: a control)
        n = int(rng.integers(1, 6))
        reads = [some_read(rng, int(rng.integers(0, rng.choice([k + s + 50, 3 * (k + s), 120_000])))) for _ in range(n)]
        buf, offs = pack(reads)
        prior = rng.integers(0, 2**32, (n, s), dtype=np.uint32)
        want = orc.mash_sketch_batch(buf, offs, k, s, out=prior.copy())
        got = mash.sketch_batch_packed(buf, offs, k, s, out=prior.copy())
        assert (got == want).all(), ("wide", it, k, s, [len(r) for r in reads])
        stats["wide"] += 1
    else:
        k, s = int(rng.choice([3, 17, 21])), int(rng.integers(0, 2))
        reads = [some_read(rng, int(rng.integers(0, 300))) for _ in range(int(rng.integers(1, 30)))]
        buf, offs = pack(reads)
        prior = rng.integers(1, 2**32, (len(reads), s), dtype=np.uint32)
        first_panic, want = None, prior.copy()
        for i, q in enumerate(reads):
            m = orc.Mash(k, s)
            m.Sketches[:] = prior[i]
            try:
                m.Sketch(q)
                want[i] = m.Sketches
            except orc.GoPanic:
                if first_panic is None:
                    first_panic = i
        try:
            got = mash.sketch_batch_packed(buf, offs, k, s, out=prior.copy())
            assert first_panic is None, ("tiny_s: no panic reported", it, k, s, first_panic)
            assert (got == want).all(), ("tiny_s", it, k, s)
        except _lib.GoPanic as e:
            assert first_panic is not None and f"sequence {first_panic} " in str(e), ("tiny_s panic", it, k, s, first_panic, str(e))
        stats["tiny_s"] += 1
    # ---- K2: compact items / 8-byte items / parts / stripes on small-valued family sketches
    bits = int(rng.choice([14, 17, 20, 24, 30]))
    s_ = int(rng.choice([16, 100, 300, 1100]))
    nfam, copies = int(rng.integers(1, 40)), int(rng.integers(1, 30))
    rows = []
    for _ in range(nfam):
        base = rng.integers(0, 1 << bits, s_, dtype=np.uint32)
        for _ in range(copies):
            m = base.copy()
            hit = rng.random(s_) < 0.15
            m[hit] = rng.integers(0, 1 << bits, int(hit.sum()), dtype=np.uint32)
            if rng.random() < 0.1:
                m[: int(rng.integers(2, 6))] = m[0]      # a hash repeated a few times inside one sketch
            m.sort()
            rows.append(m)
    Y = np.stack(rows)
    if rng.random() < 0.3:
        Y[int(rng.integers(0, len(Y)))] = Y[int(rng.integers(0, len(Y)))][::-1]          # an irregular sketch
    ny = len(Y)
    X = Y[rng.choice(ny, min(ny, 24), replace=False)]
    mode = int(rng.integers(0, 4))
    env = {}
    if mode == 1:
        env["POLYHIP_K2_COMPACT"] = "0"
    if mode == 2:
        env["POLYHIP_K2_MAX_ITEMS"] = str(s_ * max(1, ny // 3) + 1)
    os.environ.update(env)
    Yt = torch.from_numpy(Y.view(np.int32)).to(dev)
    Xt = torch.from_numpy(np.ascontiguousarray(X).view(np.int32)).to(dev)
    ct = torch.full((len(X), ny), -1, dtype=torch.int16, device=dev)
    work = torch.zeros(mash.shared_counts_workspace_bytes(len(X), s_, ny, s_), dtype=torch.uint8, device=dev)
    if mode == 3 and ny * s_ >= 64:
        nparts = int(rng.integers(2, 6))
        for p in range(nparts):
            mash.index_build_part_dev(Yt, p, nparts, work)
        mash.index_finalize_dev(ny, s_, work)
        mash.shared_counts_reuse_dev(Xt, Yt, ct, work)
        stats["parts"] += 1
    else:
        mash.shared_counts_dev(Xt, Yt, ct, work)
        stats["stripes" if mode == 2 else "compact"] += 1
    torch.cuda.synchronize()
    if mode != 2:
        stats["wide_items" if mash.index_item_bytes(work) == 8 else "compact"] += 0 if mode == 3 else 0
    for kx in env:
        os.environ.pop(kx)
    got = ct.cpu().numpy().view(np.uint16)
    for i in range(len(X)):
        for j in rng.choice(ny, min(ny, 60), replace=False):
            assert int(got[i, j]) == orc.mash_shared(X[i], Y[j]), ("k2", it, mode, bits, s_, ny, i, int(j))
    # ---- FASTA: random line widths incl. very long and one-byte lines
    parts = []
    for r in range(int(rng.integers(0, 40))):
        L = int(rng.choice([0, 1, 5, 70, 300, 5000, 20000]))
        body = bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8))
        w = int(rng.choice([1, 7, 60, 80, 4096, 100000]))
        hdr = b">" + bytes(rng.choice(list(b"abc d"), int(rng.integers(0, 40))).astype(np.uint8))
        parts.append(hdr + b"\n" + b"\n".join(body[j:j + w] for j in range(0, L, w)) + (b"\n" if L else b""))
        if rng.random() < 0.1:
            parts.append(b";comment\n")
        if rng.random() < 0.1:
            parts.append(b"\n")
    data = b"".join(parts)
    if data and rng.random() < 0.3:
        data = data[:-1]
    want, code = fr.parse_all(data)
    seqs, foffs, rec, err = fasta.pack(bytes(data))
    gotrec = [seqs[int(foffs[i]): int(foffs[i + 1])].tobytes() for i in range(len(foffs) - 1)]
    assert gotrec == [w[1] for w in want] and (err is None) == (code == 0), ("fasta", it, len(data))
    stats["fasta"] += 1
    # ---- K5: exact tandem repeats (the search runs on one block), nested periods, near-periodic
    seqs5 = []
    for _ in range(10):
        u = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 70))).astype(np.uint8))
        reps = int(rng.integers(1, 60))
        q = u * reps
        kind = int(rng.integers(0, 4))
        if kind == 1:
            q = (u * 3 + u[:1]) * reps                      # a longer block that repeats
        elif kind == 2 and len(q) > 3:
            q = q[:-int(rng.integers(1, min(len(u), len(q) - 1) + 1))]  # does not close on itself
        elif kind == 3 and len(q) > 2:
            qa = bytearray(q)
            qa[int(rng.integers(0, len(qa)))] = ord("T")
            q = bytes(qa)
        seqs5.append(q)
    buf5, offs5 = pack(seqs5)
    rot, out = seqhash.least_rotation_batch_packed(buf5, offs5, True)
    for i, q in enumerate(seqs5):
        assert int(rot[i]) == orc.booth_least_rotation(q) and out[int(offs5[i]):int(offs5[i + 1])].tobytes() == orc.rotate_sequence(q), ("k5", it, i, len(q))
    stats["k5"] += 1
    if it % 10 == 0:
        print(f"it {it}: ok {stats}", flush=True)
print("fuzz done", it, "iterations", stats)
