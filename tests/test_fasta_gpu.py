"""FASTA read feeder parity: poly_amd.fasta's device packer vs the CPU restatement of io/fasta
(*Parser).ParseNext / ParseAll (oracle/fasta_ref.py) -- same records, names, Sequence bytes and error.

Mirrors io/fasta/fasta_test.go:133-241 and example_test.go:18-36."""
import os

import numpy as np
import pytest

from oracle import fasta_ref as fr

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "fasta")


@pytest.fixture(autouse=True, params=["stream", "lines"])
def gather_kind(request, monkeypatch):
    """every test runs with the output-driven gather (a wave writes the contiguous output range of 64 lines with aligned
    dword stores: the default) and with POLYHIP_FASTA_STREAM=0 (one 8-lane group per line)"""
    if request.param == "lines":
        monkeypatch.setenv("POLYHIP_FASTA_STREAM", "0")
    else:
        monkeypatch.delenv("POLYHIP_FASTA_STREAM", raising=False)
    return request.param


def _check(data: bytes):
    from poly_amd import fasta
    want, code = fr.parse_all(data)
    raw = bytes(data)
    seqs, offs, rec, err = fasta.pack(raw)
    got = []
    for i in range(len(offs) - 1):
        start = int(rec[i])
        end = raw.index(b"\n", start)
        got.append((raw[start + 1:end], seqs[int(offs[i]): int(offs[i + 1])].tobytes()))
    assert got == want, (data[:80], got[:3], want[:3])
    assert (err is None) == (code == 0), (data[:80], err, code)
    if code:
        assert str(err) == fasta.ERRORS[code]


def test_reference_cases():
    from poly_amd import fasta
    _check(b">humen\nGATTACA\nCATGAT")            # fasta_test.go:139-142: EOF-ended fasta not valid
    _check(b">humen\nGATTACA\nCATGAT\n")
    _check(b">doggy or something\nGATTACA\n\nCATGAT\n>homunculus\nAAAA\n")
    _check(b"testing\natagtagtagtagtagatgatgatgatgagatg\n\n\n\n\n\n\n\n\n\n\n")   # :206-215
    _check(b">OK Fasta\nABGABA\n>NotOKFasta\n")    # :233-241
    base = open(os.path.join(GOLD, "base.fasta"), "rb").read()
    _check(base)
    r = fasta.records(base)
    assert r[0][0] == b"gi|5524211|gb|AAD44166.1| cytochrome b [Elephas maximus maximus]"   # example_test.go:25-30
    assert r[1][1].startswith(b"ADQLTEEQIAEFKEAFSLFDKDGDGTITTKELGTVMRSLGQNPTEAELQDMINEVDADGNGTID") and r[1][1].endswith(b"FVQMMTAK*")
    with pytest.raises(fasta.FastaError):
        fasta.records(b">OK Fasta\nABGABA\n>NotOKFasta\n")


def test_quirks_and_random_files():
    cases = [b"", b"\n", b"A", b">", b">a", b">a\n", b">a\nA", b">a\nAC", b">a\nAC\n>b", b">a\n>b", b">a\n>b\n", b">a\n>b\n>c\nAC\n",
             b">a\nAC\n>b\n>c\nGG\n", b">a\n\n>b\nAC\n", b";c\n>a\n;x\nAC\n;y\nGT\n", b"junk\n>a\nAC\n", b">a\r\nAC\r\n",
             b">a\nAC\n;tail", b">a\nAC\nG", b">a\nAC\n>", b">a\nAC\n>b\n>c", b">a\n>b\n>c", b">>\n>\nA\n", b"AC\nGT", b"AC\nGT\n",
             b">a\nAC\n\n\n", b">a\n;only\n>b\nAC\n", b">a\nAC\n>b\n;x\n", b">a\nAC\n>b\nG"]
    for c in cases:
        _check(c)
    rng = np.random.default_rng(4)
    pieces = [b">", b">id", b";c", b"", b"ACGT", b"A", b">x y", b"GG>"]
    for _ in range(400):
        n = int(rng.integers(0, 14))
        lines = [pieces[int(rng.integers(0, len(pieces)))] for _ in range(n)]
        data = b"\n".join(lines) + (b"\n" if rng.random() < 0.6 and n else b"")
        _check(data)
    # a realistic multi-line file
    recs = []
    for i in range(200):
        seq = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 900))).astype(np.uint8))
        recs.append(b">seq%d desc\n" % i + b"\n".join(seq[j:j + 70] for j in range(0, len(seq), 70)) + b"\n")
    _check(b"".join(recs))


def test_lines_longer_than_a_window_and_stretches_without_sequence():
    """single-line records of tens of kilobytes, long header / comment stretches (nothing kept for kilobytes), one-byte
    lines (thousands of lines per 4 KB), lines of 4095 / 4096 / 4097 bytes, and a dropped tail"""
    rng = np.random.default_rng(77)
    big = bytes(rng.choice(list(b"ACGT"), 50_000).astype(np.uint8))
    parts = [b">long one\n" + big + b"\n", b">" + b"h" * 9000 + b"\n" + big[:13_001] + b"\n", b";" + b"c" * 10_000 + b"\n",
             b">tiny\n" + b"\n".join(bytes([c]) for c in big[:6000]) + b"\n", b">x\n" + big[:4095] + b"\n>y\n" + big[:4096] + b"\n",
             b">z\n" + big[:4097] + b"\n"]
    data = b"".join(parts)
    _check(data)
    _check(data + b">dropped\n" + big[:9000])      # unterminated last line: the record is dropped
    _check(b"junk before\n" * 700 + data)


def test_sequence_like_lines_far_in_front_of_the_first_header():
    """lines before the first header carry no sequence (`hrank == 0`); the two per-line scans run as ONE pass over segments of
    whole 1024-line chunks, which learn where the first header is from each other: thousands of junk lines in front of it
    (segments without any header), the header exactly on a segment edge, no header at all, a header only at the very end"""
    rng = np.random.default_rng(9)
    body = b"".join(b">r%d\n" % i + bytes(rng.choice(list(b"ACGT"), 150).astype(np.uint8)) + b"\n" for i in range(3000))
    for njunk in (1023, 1024, 1025, 2048, 5000, 20_000):
        _check(b"ACGTACGT\n" * njunk + body)
    _check(b"ACGTACGT\n" * 9000)                       # no header anywhere
    _check(b"ACGTACGT\n" * 9000 + b">only\nAC\n")      # the only header at the very end
    _check(b";c\n" * 3000 + b"\n" * 3000 + body)       # comments and empty lines in front


def test_large_file_goes_through_the_multi_workgroup_scan():
    """a ~1 MB FASTA image (tens of thousands of lines: past the single-workgroup scan's limit) with ragged line
    widths, comment lines and empty lines: same records as the restated parser"""
    rng = np.random.default_rng(31)
    parts = []
    for i in range(2500):
        L = int(rng.integers(1, 700))
        body = bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8))
        w = int(rng.integers(20, 90))
        parts.append(b">rec%d some words\n" % i + b"\n".join(body[j:j + w] for j in range(0, L, w)) + b"\n")
        if i % 97 == 0:
            parts.append(b";a comment line\n")
        if i % 131 == 0:
            parts.append(b"\n")
    data = b"".join(parts)
    assert len(data) > 600_000
    _check(data)
    _check(data[:-1])      # EOF right behind the last sequence line


def test_device_image_at_any_alignment():
    """the packers read the file 16 bytes per lane and copy whole dwords: an image that starts at any byte offset of
    a device buffer, with ragged line widths (every source / destination alignment in the gather), parses the same"""
    import torch
    from poly_amd import fasta
    rng = np.random.default_rng(41)
    parts = []
    for i in range(3000):
        L = int(rng.integers(1, 400))
        body = bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8))
        w = int(rng.integers(1, 90))
        parts.append(b">r%d\n" % i + b"\n".join(body[j:j + w] for j in range(0, L, w)) + b"\n")
    data = b"".join(parts)
    want, code = fr.parse_all(data)
    assert code == 0
    wbuf = b"".join(w[1] for w in want)
    dev = torch.device("cuda:0")
    for shift in (0, 1, 3, 4, 7, 8, 13, 15):
        big = torch.zeros(len(data) + 64, dtype=torch.uint8, device=dev)
        img = big[shift:shift + len(data)]
        img.copy_(torch.from_numpy(np.frombuffer(data, np.uint8).copy()))
        nb = img.numel()
        seqs = torch.zeros(nb + 3, dtype=torch.uint8, device=dev)[3 - (shift % 4):][:nb]   # the output misaligned too
        offs = torch.zeros(nb // 2 + 4, dtype=torch.int64, device=dev)
        res = torch.zeros(4, dtype=torch.int64, device=dev)
        work = torch.empty(fasta.workspace_bytes(nb), dtype=torch.uint8, device=dev)
        fasta.pack_dev(img, seqs, offs, None, res, work)
        n, c, total, _ = (int(x) for x in res.cpu())
        assert (n, c, total) == (len(want), 0, len(wbuf)), shift
        assert seqs[:total].cpu().numpy().tobytes() == wbuf, shift
        o = offs[:n + 1].cpu().numpy()
        assert (np.diff(o) == [len(w[1]) for w in want]).all(), shift
