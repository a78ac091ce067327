"""Read feeder parity: poly_amd.fastq's device packer vs the CPU restatement of
io/fastq (*Parser).ParseNext / ParseAll (oracle/fastq_ref.py), on the reference's own fixtures
(io/fastq/data/*.fastq -> tests/golden/fastq/) and on synthetic files: same records, same Sequence bytes,
same error condition and line; then the packed batch goes straight into K1.

Mirrors io/fastq/fastq_test.go:59-66 and example_test.go:16-66."""
import glob
import os

import numpy as np
import pytest

import oracle as orc
from oracle import fastq_ref as fr

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "fastq")


def _check(data: bytes):
    from poly_amd import fastq
    seqs, offs, rec, err = fastq.pack(data)
    want, code, line = fr.parse_all(data)
    got = [seqs[int(offs[i]): int(offs[i + 1])].tobytes() for i in range(len(offs) - 1)]
    assert got == [w[1] for w in want]
    for i, w in enumerate(want):  # the identifier can be sliced from the image at rec_start
        start = int(rec[i])
        assert data[start + 1: start + 1 + len(w[0])] == w[0] or data[start:start + 1] != b"@"
    if code == 0:
        assert err is None
    else:
        assert err is not None and fastq.ERRORS[code].format(line=line) == str(err), (code, line, err)


def test_reference_fixtures():
    files = sorted(glob.glob(os.path.join(GOLD, "*.fastq")))
    assert len(files) == 7
    for f in files:
        _check(open(f, "rb").read())
    from poly_amd import fastq
    s = fastq.sequences(open(os.path.join(GOLD, "nanosavseq.fastq"), "rb").read())
    assert len(s) == 4 and s[0].startswith(b"GATGTGCGCCGTTCCAGTTGCGACG")
    for name in ("noseq", "noquality", "noidentifier", "emptyseq", "noplus", "noquality2"):  # fastq_test.go:59-66
        with pytest.raises(fastq.FastqError):
            fastq.sequences(open(os.path.join(GOLD, f"nanosavseq_{name}.fastq"), "rb").read())


def _record(rng, i, L):
    seq = bytes(rng.choice(list(b"ACGTN"), L).astype(np.uint8))
    qual = bytes(rng.integers(33, 74, L, dtype=np.uint8))
    return b"@read%d ch=%d start=%d\n" % (i, i % 7, i * 3) + seq + b"\n+\n" + qual + b"\n"


def test_synthetic_and_malformed():
    rng = np.random.default_rng(8)
    good = b"".join(_record(rng, i, int(rng.integers(1, 400))) for i in range(300))
    _check(good)
    _check(b"")
    _check(b"\n")
    _check(good[:-1])                       # last line without newline: the reference drops that record
    _check(good + b"@tail\nACGT\n")        # EOF inside a record
    _check(good + b"@tail\nACGT\n+\n")
    _check(good + b"@tail")
    _check(good[:5000] + b"\n" + good[5000:])   # a stray empty line shifts the records
    _check(good + b"@x desc\nACGT\n+\nIIII\n" + good)   # identifier field without '=': the reference panics
    _check(good + b"read\nACGT\n+\nIIII\n" + good)      # no '@'
    _check(good + b"@r\n\n+\nIIII\n" + good)            # empty sequence
    _check(good + b"@r\nACGT\n+\n\n" + good)            # empty quality
    _check(b"@r\r\nACGT\r\n+\r\nIIII\r\n")           # CR is kept, as in the reference
    _check(b"@a k=v  \nAC\n+\nII\n")                   # empty field after a double space


def test_shortest_records():
    """the reference never looks at the third line (fastq.go:182 discards it), so a record can be 7 bytes:
    more records than nbytes / 8, single- and multi-workgroup scan sizes"""
    from poly_amd import fastq
    rec = b"@\nA\n\nI\n"
    for count in (1, 3, 5000, 300_000):
        data = rec * count
        seqs, offs, starts, err = fastq.pack(data)
        assert err is None and len(offs) == count + 1 and bytes(seqs) == b"A" * count
        assert (starts == np.arange(count, dtype=np.uint64) * 7).all()
    _check(rec * 40 + b"@r\nAC\n+\nII\n" + rec * 3)
    _check(rec * 9 + b"@\n\n\nI\n")


def test_max_records_bounds_every_write():
    """caller buffers sized for max_records + 1 entries: nothing is written past them (sentinels survive), code 7"""
    import torch
    from poly_amd import fastq
    rng = np.random.default_rng(10)
    data = b"".join(_record(rng, i, 50) for i in range(2000))
    dev = torch.device("cuda:0")
    img = torch.from_numpy(np.frombuffer(data, np.uint8).copy()).to(dev)
    cap = 100
    seqs = torch.empty(img.numel(), dtype=torch.uint8, device=dev)
    offs = torch.full((cap + 1 + 64,), -7, dtype=torch.int64, device=dev)
    rec = torch.full((cap + 1 + 64,), -7, dtype=torch.int64, device=dev)
    res = torch.zeros(4, dtype=torch.int64, device=dev)
    work = torch.empty(fastq.workspace_bytes(img.numel()), dtype=torch.uint8, device=dev)
    fastq.pack_dev(img, seqs, offs, rec, res, work, max_records=cap)
    n, code, _, total = (int(x) for x in res.cpu())
    assert (n, code, total) == (cap, 7, 50 * cap)
    assert (offs[cap + 1:] == -7).all() and (rec[cap:] == -7).all()
    assert offs[:cap + 1].tolist() == [50 * i for i in range(cap + 1)]
    seqs_h, offs_h, _, err = fastq.pack(data, max_records=cap)
    assert len(offs_h) == cap + 1 and "more records" in str(err)


def test_packed_batch_feeds_the_sketch_kernel():
    """file image -> device packer -> K1, no host parse: equals sketching the oracle's records"""
    import torch
    from poly_amd import fastq, mash
    rng = np.random.default_rng(9)
    data = b"".join(_record(rng, i, int(rng.integers(900, 1400))) for i in range(500))
    dev = torch.device("cuda:0")
    img = torch.from_numpy(np.frombuffer(data, np.uint8).copy()).to(dev)
    nb = img.numel()
    seqs = torch.empty(nb, dtype=torch.uint8, device=dev)
    offs = torch.zeros(nb // 7 + 2, dtype=torch.int64, device=dev)
    res = torch.zeros(4, dtype=torch.int64, device=dev)
    work = torch.empty(fastq.workspace_bytes(nb), dtype=torch.uint8, device=dev)
    fastq.pack_dev(img, seqs, offs, None, res, work)
    n, code, _, total = (int(x) for x in res.cpu())
    assert (n, code) == (500, 0)
    sk = torch.zeros((n, 200), dtype=torch.int32, device=dev)
    mash.sketch_batch_dev(seqs, offs[: n + 1], 21, 200, sk)
    torch.cuda.synchronize()
    want_recs, _, _ = fr.parse_all(data)
    buf = np.frombuffer(b"".join(r[1] for r in want_recs), np.uint8)
    o = np.zeros(n + 1, np.uint64)
    o[1:] = np.cumsum([len(r[1]) for r in want_recs])
    assert total == len(buf)
    assert (sk.cpu().numpy().view(np.uint32) == orc.mash_sketch_batch(buf, o, 21, 200)).all()


def test_large_file_goes_through_the_multi_workgroup_scan():
    """a 2.5 MB FASTQ image (more records than the single-workgroup scan takes): same records as the restated parser,
    also with a malformed record deep inside"""
    rng = np.random.default_rng(32)
    good = b"".join(_record(rng, i, int(rng.integers(1, 120))) for i in range(24000))
    assert len(good) > 2_200_000
    _check(good)
    _check(good[:1_500_000] + b"@r\nACGT\n+\n\n" + good[1_500_000:])


def test_device_image_at_any_alignment():
    """the image may start at any byte of a device buffer (16-byte loads on the aligned address space, the bytes in
    front masked out) and the packed sequences may land at any alignment (dword copies with byte ends)"""
    import torch
    from poly_amd import fastq
    rng = np.random.default_rng(42)
    data = b"".join(_record(rng, i, int(rng.integers(1, 300))) for i in range(4000))
    want, code, _ = fr.parse_all(data)
    assert code == 0
    wbuf = b"".join(w[1] for w in want)
    dev = torch.device("cuda:0")
    for shift in (0, 1, 2, 5, 8, 11, 15):
        big = torch.zeros(len(data) + 64, dtype=torch.uint8, device=dev)
        img = big[shift:shift + len(data)]
        img.copy_(torch.from_numpy(np.frombuffer(data, np.uint8).copy()))
        nb = img.numel()
        seqs = torch.zeros(nb + 3, dtype=torch.uint8, device=dev)[(shift + 1) % 4:][:nb]
        offs = torch.zeros(nb // 7 + 2, dtype=torch.int64, device=dev)
        res = torch.zeros(4, dtype=torch.int64, device=dev)
        work = torch.empty(fastq.workspace_bytes(nb), dtype=torch.uint8, device=dev)
        fastq.pack_dev(img, seqs, offs, None, res, work)
        n, c, _, total = (int(x) for x in res.cpu())
        assert (n, c, total) == (len(want), 0, len(wbuf)), shift
        assert seqs[:total].cpu().numpy().tobytes() == wbuf, shift
