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
