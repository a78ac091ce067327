"""CPU restatement of clone's ligation + seqhash dedup (clone/clone.go:135-353) -- TEST INFRASTRUCTURE ONLY.

Follows the reference statement by statement, with one seqhash.Hash (the oracle's restatement, poly_oracle.c
orc_seqhash) per candidate construct INSIDE the recursion, exactly where clone.go:275 and :305 call it.  The
product (poly_amd/clone.py) hoists those calls into one batched device call; tests compare the two.
Pinned by tests/test_oracle_golden.py on clone/example_test.go:11-31 (ExampleGoldenGate's printed rotation).
"""
from __future__ import annotations

import re

import oracle as orc


class Fragment:
    def __init__(self, Sequence, ForwardOverhang, ReverseOverhang):
        self.Sequence, self.ForwardOverhang, self.ReverseOverhang = Sequence, ForwardOverhang, ReverseOverhang

    def key(self):
        return (self.Sequence, self.ForwardOverhang, self.ReverseOverhang)


ENZYMES = {  # clone.go:356-362: name -> (forward regexp, reverse regexp, skip, overhead length, recognition site)
    "BsaI": ("GGTCTC", "GAGACC", 1, 4, "GGTCTC"),
    "BbsI": ("GAAGAC", "GTCTTC", 2, 4, "GAAGAC"),
    "BtgZI": ("GCGATG", "CATCGC", 10, 4, "GCGATG"),
}


def _rc(s: str) -> str:
    return orc.reverse_complement(s).decode("latin-1")


def _hash(construct: str, circular: bool) -> str:
    try:
        return orc.seqhash(construct, "DNA", circular, True)
    except orc.SeqhashError:
        return ""  # the reference drops Hash's error and uses the empty string (clone.go:275)


def cut_with_enzyme(sequence: str, circular: bool, directional: bool, name: str):
    """clone.go:135-268"""
    re_for, re_rev, skip, overhead, site = ENZYMES[name]
    part_len = len(sequence)
    if circular:
        sequence = (sequence + sequence).upper()
    else:
        sequence = sequence.upper()
    palindromic = site == _rc(site)
    forward_overhangs, reverse_overhangs = [], []
    for m in re.finditer(re_for, sequence):
        forward_overhangs.append({"Position": m.end() + skip, "Forward": True})
    if not palindromic:
        for m in re.finditer(re_rev, sequence):
            reverse_overhangs.append({"Position": m.start() - skip, "Forward": False})
    overhangs = []
    for overhang_set in (forward_overhangs, reverse_overhangs):
        if len(overhang_set) > 0:
            if not circular and overhang_set[-1]["Position"] + skip + overhead > len(sequence):
                overhang_set = overhang_set[:-1]
        overhangs.extend(overhang_set)
    overhangs.sort(key=lambda o: o["Position"])
    fragments = []
    if len(overhangs) == 1 and not directional and not circular:
        p = overhangs[0]["Position"]
        if len(forward_overhangs) > 0:
            fragments.append(Fragment(sequence[p + overhead:], sequence[p:p + overhead], ""))
            fragments.append(Fragment(sequence[:p], "", sequence[p:p + overhead]))
        else:
            fragments.append(Fragment(sequence[:p - overhead], "", sequence[p - overhead:p]))
            fragments.append(Fragment(sequence[p:], sequence[p - overhead:p], ""))
        return fragments
    if len(overhangs) == 2 and not directional and circular:
        p = overhangs[0]["Position"]
        fragments.append(Fragment(sequence[p + overhead:part_len] + sequence[:p], sequence[p:p + overhead],
                                  sequence[p:p + overhead]))
        return fragments
    fragment_sequences = []
    if len(overhangs) > 1:
        for i in range(len(overhangs) - 1):
            cur, nxt = overhangs[i], overhangs[i + 1]
            if directional and not palindromic:
                if cur["Forward"] and not nxt["Forward"]:
                    fragment_sequences.append(sequence[cur["Position"]:nxt["Position"]])
                if nxt["Position"] - (len(site) + skip) > part_len:
                    break
            else:
                fragment_sequences.append(sequence[cur["Position"]:nxt["Position"]])
                if nxt["Position"] - (len(site) + skip) > part_len:
                    break
        for fs in fragment_sequences:
            if len(fs) > 8:
                fragments.append(Fragment(fs[overhead:len(fs) - overhead], fs[:overhead], fs[len(fs) - overhead:]))
    return fragments


def recurse_ligate(seed, fragment_list, used_fragments, existing):
    """clone.go:269-318"""
    if seed.ForwardOverhang == seed.ReverseOverhang:
        construct = seed.ForwardOverhang + seed.Sequence
        h = _hash(construct, True)
        if h in existing:
            return [], []
        existing.add(h)
        return [construct], []
    open_constructs, infinite_constructs = [], []
    for new in fragment_list:
        new_seed, attached = None, False
        if seed.ReverseOverhang == new.ForwardOverhang:
            attached = True
            new_seed = Fragment(seed.Sequence + seed.ReverseOverhang + new.Sequence, seed.ForwardOverhang, new.ReverseOverhang)
        if seed.ReverseOverhang == _rc(new.ReverseOverhang) and seed.ReverseOverhang != _rc(seed.ReverseOverhang):
            attached = True
            new_seed = Fragment(seed.Sequence + seed.ReverseOverhang + _rc(new.Sequence), seed.ForwardOverhang,
                                _rc(new.ForwardOverhang))
        if attached:
            for used in used_fragments:
                if used.Sequence == new.Sequence:
                    infinite = used.ForwardOverhang + used.Sequence + used.ReverseOverhang
                    h = _hash(infinite, False)
                    if h in existing:
                        return [], []
                    existing.add(h)
                    return [], [infinite]
            # `usedFragments = append(usedFragments, newFragment)` (:314) assigns the function's own variable, so the
            # list keeps growing across the SIBLINGS of this loop, not only down the recursion
            used_fragments = used_fragments + [new]
            o, i = recurse_ligate(new_seed, fragment_list, used_fragments, existing)
            open_constructs += o
            infinite_constructs += i
    return open_constructs, infinite_constructs


def circular_ligate(fragments):
    """clone.go:321-335"""
    out, inf, existing = [], [], set()
    for f in fragments:
        o, i = recurse_ligate(f, fragments, [], existing)
        out += o
        inf += i
    return out, inf


def golden_gate(parts, enzyme: str):
    """clone.go:345-353; parts = [(sequence, circular)]"""
    fragments = []
    for seq, circular in parts:
        fragments += cut_with_enzyme(seq, circular, True, enzyme)
    return circular_ligate(fragments)
