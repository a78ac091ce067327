"""CPU restatement of io/fasta (*Parser).ParseNext / ParseAll (io/fasta/fasta.go:102-238).

TEST INFRASTRUCTURE ONLY (see oracle/poly_oracle.h): the checker for poly_amd.fasta's device feeder.
Pinned by tests/test_oracle_golden.py on the reference's own tests: fasta_test.go:133-172 (TestParser),
:206-215 (TestReadEmptyFasta), :233-241 (TestParseEOFAfterName), example_test.go:18-36,100-114
(data/base.fasta -> tests/golden/fasta/base.fasta).  Pure-Python loops: fine for KB-sized cases.
"""
from __future__ import annotations


def parse_all(data: bytes):
    """-> (records [(name, sequence)], error_code): 0 none, 1 "did not find fasta start '>'" (:223),
    2 "empty fasta sequence" (:227).  Like ParseN (:105-117), records before the error are kept, an
    EOF-ended last record is dropped and an error that wraps io.EOF is not an error."""
    n, pos, out = len(data), 0, []
    while True:
        if pos >= n:                       # Peek(1) fails: EOF, clean end (:150-153)
            return out, 0
        looking, name, seq, eof_err = True, b"", bytearray(), False
        while True:
            i = data.find(b"\n", pos)
            if i < 0:
                line, pos, at_eof = data[pos:], n, True
            else:
                line, pos, at_eof = data[pos:i + 1], i + 1, False
            skippable = len(line) <= 1 or line[0:1] == b";"      # :169
            if at_eof:                                            # :174-193
                if skippable:
                    eof_err = False
                else:
                    seq += line
                    eof_err = True
                break
            line = line[:-1]
            nxt = data[pos:pos + 1]
            if not looking and nxt == b">":                       # :197-204
                if not skippable:
                    seq += line
                break
            elif skippable:
                continue
            if looking:                                           # :208-216
                if line[0:1] == b">":
                    name = line[1:]
                    looking = False
                continue
            seq += line                                           # :219
        if looking:                                               # :223 (wraps err: EOF -> not an error)
            return out, (0 if eof_err else 1)
        if len(seq) == 0:                                         # :226-229
            return out, (0 if eof_err else 2)
        if eof_err:                                               # returned with EOF: ParseN drops it
            return out, 0
        out.append((bytes(name), bytes(seq)))
