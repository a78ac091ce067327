"""primers/pcr primer design of bebop/poly on top of the batched Tm kernel.

Mirrors primers/pcr/pcr.go:44-66: ``DesignPrimersWithOverhangs`` / ``DesignPrimers`` grow a primer from
15 nt until ``primers.MeltingTemp`` reaches the target (pcr.go:47-53).  Here every candidate length of
every gene is scored by ONE polyhip_santalucia_batch call (the grow loop becomes a lookup), which is
what makes "design primers for every CDS of a genome" (tutorials/002_primer_design_test.go:82-99) a
single device call.  ``SimulateSimple`` / ``Simulate`` (pcr.go:74-188) get every primer's minimal binding
length from ONE batched Tm call over all of its suffixes (the :96-101 loop becomes a lookup); the
binding-site search and fragment assembly stay host orchestration, as in the reference.
"""
from __future__ import annotations

import numpy as np

from . import _lib, primers
from .mash import _pack

minimalPrimerLength = 7  # pcr.go:35
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


# ---- pcr.Simulate (pcr.go:74-203) --------------------------------------------------------------
def _minimal_lengths(primer_list, targetTm: float):
    """pcr.go:95-101 for every primer: the LAST suffix length (from 7 up) whose MeltingTemp is still below
    the target, 0 if already the 7-mer reaches it, len(primer) if the whole primer stays below (the
    primer is then rejected, :104).  Suffixes are scored in growing windows (7..70, then x4 while some primer
    has not reached the target): one polyhip_santalucia_batch call per window, and a 100 kb amplicon used as a
    primer in Simulate's second round (:181) costs the few dozen suffixes the reference would have scored."""
    for primer in primer_list:
        if len(primer) < minimalPrimerLength:
            raise _lib.GoPanic(_lib.ERR_PANIC, "slice bounds out of range (primer shorter than 7 nt, pcr.go:96)")
    minimal = [0] * len(primer_list)
    open_ = list(range(len(primer_list)))
    lo, span = minimalPrimerLength, 64
    while open_:
        cands, owner = [], []
        for i in open_:
            primer = primer_list[i]
            for index in range(lo, min(lo + span, len(primer) + 1)):
                cands.append(primer[len(primer) - index:])
                owner.append((i, index))
        if not cands:
            break
        tm, _, _ = primers.santalucia_batch_packed(*_pack(cands), 500e-9, 50e-3, 0.0)
        reached = set()
        for (i, index), t in zip(owner, tm):
            if i in reached:
                continue
            if not (t < targetTm):
                reached.add(i)
                continue
            minimal[i] = index
        open_ = [i for i in open_ if i not in reached and lo + span <= len(primer_list[i])]
        lo += span
        span *= 4
    return minimal


def _lookup(sequence: bytes, pattern: bytes):
    """suffixarray.Lookup(pattern, -1) (pcr.go:108,111): all, also overlapping, occurrences; none for an empty pattern"""
    if not pattern:
        return []
    out, at = [], sequence.find(pattern)
    while at >= 0:
        out.append(at)
        at = sequence.find(pattern, at + 1)
    return out


def _fragments(sequence, fwd_loc, rev_loc, fwd_idx, rev_idx, minimal_primers, primer_list):
    """generatePcrFragments, pcr.go:190-203"""
    out = []
    for fi in fwd_idx:
        full_fwd = primer_list[fi]
        head = full_fwd[:len(full_fwd) - len(minimal_primers[fi])]
        for ri in rev_idx:
            out.append(head + sequence[fwd_loc:rev_loc] + _revcomp(primer_list[ri]))
    return out


def SimulateSimple(sequences, targetTm: float, circular: bool, primerList):
    """pcr.go:74-165.  ``primerList`` is upper-cased in place like the Go slice (:76-78)."""
    for i in range(len(primerList)):
        primerList[i] = primerList[i].upper()
    plist = [p.encode("latin-1") for p in primerList]
    minimal_len = _minimal_lengths(plist, targetTm) if len(sequences) else []
    fragments = []
    for sequence in sequences:
        sequence = sequence.upper().encode("latin-1")
        fwd_locs, rev_locs = {}, {}
        minimal_primers = [b""] * len(plist)
        for pi, primer in enumerate(plist):
            minimal = primer[len(primer) - minimal_len[pi]:]
            if minimal == primer:  # never reaches the target Tm: rejected (:104)
                continue
            minimal_primers[pi] = minimal
            for loc in _lookup(sequence, minimal):
                fwd_locs.setdefault(loc, []).append(pi)
            for loc in _lookup(sequence, _revcomp(minimal)):
                rev_locs.setdefault(loc, []).append(pi)
        fwd_ints, rev_ints = sorted(fwd_locs), sorted(rev_locs)
        for at, fl in enumerate(fwd_ints):
            if at + 1 != len(fwd_ints):
                nxt = fwd_ints[at + 1]
                rl = next((r for r in rev_ints if fl < r < nxt), None)  # first reverse site before the next forward one
                if rl is not None:
                    fragments += _fragments(sequence, fl, rl, fwd_locs[fl], rev_locs[rl], minimal_primers, plist)
                continue
            later = [r for r in rev_ints if fl < r]
            for rl in later:
                fragments += _fragments(sequence, fl, rl, fwd_locs[fl], rev_locs[rl], minimal_primers, plist)
            if circular and not later:  # look on the other side of the origin (:150-160)
                rotated = sequence[fl:] + sequence[:fl]
                for rl in rev_ints:
                    if fwd_ints[0] > rl:
                        fragments += _fragments(rotated, 0, len(sequence) - fl + rl, fwd_locs[fl], rev_locs[rl],
                                                minimal_primers, plist)
    return [f.decode("latin-1") for f in fragments]


def Simulate(sequences, targetTm: float, circular: bool, primerList):
    """pcr.go:173-188 -> (fragments, error): error is None, "Primers are too short." (fragments None) or
    "Concatemerization detected in PCR." (with the first round's fragments)."""
    for primer in primerList:
        if len(primer) < minimalPrimerLength:
            return None, "Primers are too short."
    initial = SimulateSimple(sequences, targetTm, circular, primerList)
    subsequent = SimulateSimple(sequences, targetTm, circular, list(primerList) + initial)
    if len(initial) != len(subsequent):
        return initial, "Concatemerization detected in PCR."
    return initial, None
