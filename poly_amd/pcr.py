"""primers/pcr primer design of bebop/poly on top of the batched Tm kernel.

Mirrors primers/pcr/pcr.go:44-66: ``DesignPrimersWithOverhangs`` / ``DesignPrimers`` grow a primer from
15 nt until ``primers.MeltingTemp`` reaches the target (pcr.go:47-53).  Here every candidate length of
every gene is scored by ONE polyhip_santalucia_batch call (the grow loop becomes a lookup), which is
what makes "design primers for every CDS of a genome" (tutorials/002_primer_design_test.go:82-99) a
single device call.  ``Simulate`` (pcr.go:74-200, suffix-array lookups) is host orchestration and out of scope.
"""
from __future__ import annotations

import numpy as np

from . import _lib, primers
from .mash import _pack

designedMinimalPrimerLength = 15  # pcr.go:38

_COMP = bytes.maketrans(b"ABCDGHKMNRSTVWYabcdghkmnrstvwy", b"TVGHCDMKNYSABWRtvghcdmknysabwr")


def _revcomp(b: bytes) -> bytes:
    """transform.ReverseComplement (transform.go:15-23,78-109); unmapped bytes -> 0x00"""
    table = bytearray(256)
    for k in b"ABCDGHKMNRSTVWYabcdghkmnrstvwy":
        table[k] = bytes([k]).translate(_COMP)[0]
    return b[::-1].translate(bytes(table))


def DesignPrimersBatch(sequences, targetTm: float, max_growth: int = 64):
    """[(forward, reverse)] for every sequence: the shortest prefix / reverse-complemented suffix of at
    least 15 nt whose MeltingTemp is >= targetTm (pcr.go:47-53).  A sequence whose primer would have to
    grow past its own length raises GoPanic (the reference slices out of range there)."""
    seqs = [(s.encode("latin-1") if isinstance(s, str) else bytes(s)).upper() for s in sequences]  # pcr.go:45
    cands, owner = [], []
    for i, s in enumerate(seqs):
        if len(s) < designedMinimalPrimerLength:
            raise _lib.GoPanic(_lib.ERR_PANIC, "slice bounds out of range (sequence shorter than 15 nt, pcr.go:46)")
        top = min(len(s), designedMinimalPrimerLength + max_growth)
        for L in range(designedMinimalPrimerLength, top + 1):
            cands.append(s[:L])
            owner.append((i, 0, L))
            cands.append(_revcomp(s[len(s) - L:]))
            owner.append((i, 1, L))
    tm, _, _ = primers.santalucia_batch_packed(*_pack(cands), 500e-9, 50e-3, 0.0)  # primers.MeltingTemp defaults
    best = {}
    for (i, strand, L), t, c in zip(owner, tm, cands):
        if t >= targetTm and (i, strand) not in best:  # candidates are in increasing L
            best[(i, strand)] = c
    out = []
    for i, s in enumerate(seqs):
        if (i, 0) not in best or (i, 1) not in best:
            if designedMinimalPrimerLength + max_growth >= len(s):
                raise _lib.GoPanic(_lib.ERR_PANIC, "slice bounds out of range (no primer reaches the target Tm, pcr.go:48)")
            return DesignPrimersBatch(sequences, targetTm, max_growth * 4)
        out.append((best[(i, 0)].decode("latin-1"), best[(i, 1)].decode("latin-1")))
    return out


def DesignPrimersWithOverhangs(sequence, forwardOverhang, reverseOverhang, targetTm: float):
    """pcr.go:44-60"""
    fwd, rev = DesignPrimersBatch([sequence], targetTm)[0]
    ro = reverseOverhang.encode("latin-1") if isinstance(reverseOverhang, str) else bytes(reverseOverhang)
    return forwardOverhang + fwd, _revcomp(ro).decode("latin-1") + rev


def DesignPrimers(sequence, targetTm: float):
    """pcr.go:64-66"""
    return DesignPrimersWithOverhangs(sequence, "", "", targetTm)
