"""search/align/matrix of bebop/poly (host-side mirror of matrix.go; the canned
tables of matrices.go are data and are supplied by the caller -- only NUC_4
is included, because BASELINE config 4 and the reference's examples use it;
any other table, NUC_4_4 or the protein matrices, goes through
NewSubstitutionMatrix like a user's own)."""
from __future__ import annotations

from . import alphabet as _alphabet


class SubstitutionMatrix:
    """matrix.go:13-38"""

    def __init__(self, first: _alphabet.Alphabet, second: _alphabet.Alphabet, scores):
        self.FirstAlphabet = first
        self.SecondAlphabet = second
        self._scores = [list(r) for r in scores]

    def Score(self, a: str, b: str) -> int:
        i = self.FirstAlphabet.Encode(a)   # first alphabet checked first (matrix.go:29-32)
        j = self.SecondAlphabet.Encode(b)
        return self._scores[i][j]


def NewSubstitutionMatrix(first, second, scores) -> SubstitutionMatrix:
    """matrix.go:20-25"""
    if len(first.Symbols()) != len(scores) or len(second.Symbols()) != len(scores[0]):
        raise ValueError("invalid dimensions of substitution matrix")
    return SubstitutionMatrix(first, second, scores)


_letters = list("ABCDEFGHIJKLMNOPQRSTUVWXYZ")
#: matrix.go:40-73
Default = NewSubstitutionMatrix(_alphabet.NewAlphabet(_letters), _alphabet.NewAlphabet(_letters),
                                [[1 if i == j else -1 for j in range(26)] for i in range(26)])

#: matrices.go:33-40 (alphabet order "-ACGT")
NUC_4 = [
    [0, 0, 0, 0, 0],
    [0, 5, -4, -4, -4],
    [0, -4, 5, -4, -4],
    [0, -4, -4, 5, -4],
    [0, -4, -4, -4, 5],
]
