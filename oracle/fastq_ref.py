"""CPU restatement of io/fastq (*Parser).ParseNext / ParseAll (io/fastq/fastq.go:84-216).

TEST INFRASTRUCTURE ONLY (see oracle/poly_oracle.h): the checker for poly_amd.fastq's device feeder.
Pinned on the reference's own fixtures (tests/golden/fastq/*.fastq, copied from io/fastq/data/) by
tests/test_oracle_golden.py: nanosavseq.fastq parses, the six exception files fail, ExampleParser's identifiers.
Pure-Python loops: fine for the KB-sized cases it checks.
"""
from __future__ import annotations


class GoPanic(Exception):
    pass


def parse_all(data: bytes):
    """-> (records [(identifier, sequence, quality)], error_code, error_line).
    error_code: 0 none; 1 no '@' (:203); 2 empty sequence (:176); 3 empty quality (:197); 4 unexpected EOF (:147);
    5 / 6: the reference panics (index [0] of an empty identifier line :156, optionalSplits[1] :163)."""
    pos, line_no, out = 0, 0, []
    n = len(data)

    def read_slice():
        nonlocal pos
        i = data.find(b"\n", pos)
        if i < 0:  # ReadSlice returns what is left with io.EOF
            chunk, pos_new, eof = data[pos:], n, True
        else:
            chunk, pos_new, eof = data[pos:i + 1], i + 1, False
        pos = pos_new
        return chunk, eof

    while pos < n:  # Peek(1) at EOF ends the parse cleanly (:118-121)
        looking = True
        line, eof = read_slice()                      # identifier (:146)
        line_no += 1
        if eof:
            return out, 4, line_no + 1
        line = line[:-1]
        if len(line) == 0:
            return out, 5, line_no                    # string(line)[0] panics
        if line[0:1] == b"@":
            looking = False
        splits = line.split(b" ")
        ident = splits[0][1:]
        for opt in splits[1:]:
            if b"=" not in opt:
                return out, 6, line_no                # optionalSplits[1] panics
        line, eof = read_slice()                      # sequence (:170)
        line_no += 1
        if eof:
            return out, 4, line_no + 1
        if len(line) <= 1:
            return out, 2, line_no
        seq = line[:-1]
        _, eof = read_slice()                         # '+' (:182)
        line_no += 1
        if eof:
            return out, 4, line_no + 1
        line, eof = read_slice()                      # quality (:190)
        line_no += 1
        if eof:
            return out, 4, line_no + 1
        if len(line) <= 1:
            return out, 3, line_no
        if looking:
            return out, 1, line_no
        out.append((ident, seq, line[:-1]))
    return out, 0, 0
