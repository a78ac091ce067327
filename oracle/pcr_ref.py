"""TEST INFRASTRUCTURE ONLY -- CPU restatement of bebop/poly's primers/pcr package.

Follows /root/reference/primers/pcr/pcr.go line by line (plain Python loops; small cases only):
  * DesignPrimersWithOverhangs / DesignPrimers   pcr.go:44-66
  * SimulateSimple                               pcr.go:74-165
  * Simulate                                     pcr.go:173-188
  * generatePcrFragments                         pcr.go:190-203
on top of the C oracle's ``melting_temp`` (primers.go:121-128) and ``reverse_complement``
(transform.go:15-23).  Pinned on the reference's own tests in tests/test_oracle_golden.py
(pcr_test.go:14-95, example_test.go:36-63).  Nothing under poly_amd/ may import this module.
"""
from __future__ import annotations

from . import melting_temp, reverse_complement

minimalPrimerLength = 7            # pcr.go:35
designedMinimalPrimerLength = 15   # pcr.go:38


class GoPanic(Exception):
    """the Go code would panic (slice bounds out of range)"""


def _rc(s: str) -> str:
    return reverse_complement(s.encode("latin-1")).decode("latin-1")


def design_primers_with_overhangs(sequence: str, fwd_overhang: str, rev_overhang: str, target_tm: float):
    """pcr.go:44-60"""
    sequence = sequence.upper()                                   # :45
    n = len(sequence)
    if n < designedMinimalPrimerLength:
        raise GoPanic("slice bounds out of range")
    fwd = sequence[:designedMinimalPrimerLength]                  # :46
    add = 0
    while melting_temp(fwd) < target_tm:                          # :47-50
        add += 1
        if designedMinimalPrimerLength + add > n:
            raise GoPanic("slice bounds out of range")
        fwd = sequence[:designedMinimalPrimerLength + add]
    rev = _rc(sequence[n - designedMinimalPrimerLength:])         # :51
    add = 0
    while melting_temp(rev) < target_tm:                          # :52-55
        add += 1
        if designedMinimalPrimerLength + add > n:
            raise GoPanic("slice bounds out of range")
        rev = _rc(sequence[n - (designedMinimalPrimerLength + add):])
    return fwd_overhang + fwd, _rc(rev_overhang) + rev            # :57-59


def design_primers(sequence: str, target_tm: float):
    """pcr.go:64-66"""
    return design_primers_with_overhangs(sequence, "", "", target_tm)


def _lookup(sequence: str, pattern: str):
    """suffixarray.Index.Lookup(pattern, -1): every (overlapping) occurrence; nil for an empty pattern"""
    if not pattern:
        return []
    out, at = [], sequence.find(pattern)
    while at >= 0:
        out.append(at)
        at = sequence.find(pattern, at + 1)
    return out


def _generate_pcr_fragments(sequence, fwd_loc, rev_loc, fwd_idx, rev_idx, minimal_primers, primer_list):
    """pcr.go:190-203"""
    frags = []
    for fi in fwd_idx:
        minimal = minimal_primers[fi]
        full_fwd = primer_list[fi]
        for ri in rev_idx:
            full_rev = _rc(primer_list[ri])
            frags.append(full_fwd[:len(full_fwd) - len(minimal)] + sequence[fwd_loc:rev_loc] + full_rev)
    return frags


def simulate_simple(sequences, target_tm: float, circular: bool, primer_list):
    """pcr.go:74-165.  ``primer_list`` is upper-cased IN PLACE like the Go slice (:76-78)."""
    for i in range(len(primer_list)):
        primer_list[i] = primer_list[i].upper()
    fragments = []
    for sequence in sequences:
        sequence = sequence.upper()                               # :82
        fwd_locs, rev_locs = {}, {}
        minimal_primers = [""] * len(primer_list)
        for pi, primer in enumerate(primer_list):
            minimal_length = 0                                    # :95
            index = minimalPrimerLength
            while True:                                           # :96-101
                if index > len(primer):
                    raise GoPanic("slice bounds out of range")
                if not (melting_temp(primer[len(primer) - index:]) < target_tm):
                    break
                minimal_length = index
                if primer[len(primer) - index:] == primer:
                    break
                index += 1
            minimal = primer[len(primer) - minimal_length:]       # :103
            if minimal != primer:                                 # :104
                minimal_primers[pi] = minimal
                for loc in _lookup(sequence, minimal):            # :108-110
                    fwd_locs.setdefault(loc, []).append(pi)
                for loc in _lookup(sequence, _rc(minimal)):       # :111-113
                    rev_locs.setdefault(loc, []).append(pi)
        fwd_ints = sorted(fwd_locs)                               # :117-126
        rev_ints = sorted(rev_locs)
        for index, fl in enumerate(fwd_ints):                     # :129
            if index + 1 != len(fwd_ints):                        # :131
                for rl in rev_ints:
                    if fl < rl < fwd_ints[index + 1]:             # :134
                        fragments += _generate_pcr_fragments(sequence, fl, rl, fwd_locs[fl], rev_locs[rl],
                                                             minimal_primers, primer_list)
                        break
            else:
                found = False
                for rl in rev_ints:                               # :143-148
                    if fl < rl:
                        fragments += _generate_pcr_fragments(sequence, fl, rl, fwd_locs[fl], rev_locs[rl],
                                                             minimal_primers, primer_list)
                        found = True
                if circular and not found:                        # :150-160
                    for rl in rev_ints:
                        if fwd_ints[0] > rl:
                            rotated = sequence[fl:] + sequence[:fl]
                            fragments += _generate_pcr_fragments(rotated, 0, len(sequence[fl:]) + rl, fwd_locs[fl],
                                                                 rev_locs[rl], minimal_primers, primer_list)
    return fragments


def simulate(sequences, target_tm: float, circular: bool, primer_list):
    """pcr.go:173-188 -> (fragments, error message or None)"""
    for primer in primer_list:
        if len(primer) < minimalPrimerLength:
            return None, "Primers are too short."
    initial = simulate_simple(sequences, target_tm, circular, primer_list)
    subsequent = simulate_simple(sequences, target_tm, circular, list(primer_list) + initial)
    if len(initial) != len(subsequent):
        return initial, "Concatemerization detected in PCR."
    return initial, None
