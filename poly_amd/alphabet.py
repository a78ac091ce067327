"""alphabet/alphabet.go of bebop/poly (host-side mirror; stays on the host in
the Go drop-in too -- it is consumed by the align wrapper, not replaced)."""
from __future__ import annotations


class Error(Exception):
    """alphabet.go:14-22"""


class Alphabet:
    """alphabet.go:9-12,25-41: symbols + map symbol->index (and index->index);
    a repeated symbol keeps its LAST index."""

    def __init__(self, symbols):
        self.symbols = list(symbols)
        self.encoding = {}
        for index, symbol in enumerate(self.symbols):
            self.encoding[symbol] = index
            self.encoding[index] = index

    def Encode(self, symbol):
        if symbol not in self.encoding:
            raise Error(f"Symbol {symbol} not in alphabet")  # alphabet.go:38
        return self.encoding[symbol]

    def Decode(self, code):
        if not isinstance(code, int) or code < 0 or code >= len(self.symbols):
            raise Error(f"Code {code} not in alphabet")
        return self.symbols[code]

    def Extend(self, symbols):
        return Alphabet(self.symbols + list(symbols))

    def Symbols(self):
        return self.symbols


def NewAlphabet(symbols) -> Alphabet:
    return Alphabet(symbols)


DNA = NewAlphabet(["A", "C", "G", "T"])
RNA = NewAlphabet(["A", "C", "G", "U"])
Protein = NewAlphabet(list("ACDEFGHIKLMNPQRSTVWY"))
