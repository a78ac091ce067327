"""clone's ligation dedup on the batched seqhash kernel (clone/clone.go:269-340).

The reference's ``recurseLigate`` calls ``seqhash.Hash`` once per candidate construct (clone.go:275 for a
circularised construct, :305 for an "infinite" linear one) and keeps the construct only if its hash is new.
The hash never steers the recursion -- both call sites are leaves that return whatever the map says -- so the
recursion can run once dry to list the candidates, ALL of them be hashed by one ``polyhip_seqhash_batch`` call
per flag group (circular double-stranded, linear double-stranded), and the recursion run again on those
results: exactly the constructs, in exactly the order, the reference keeps.

Only what that path needs of clone.go is mirrored here, as host orchestration: ``Part`` / ``Fragment`` /
``Enzyme`` (:57-86), ``CutWithEnzyme`` (:135-268), ``CircularLigate`` (:321-335), ``GoldenGate`` (:345-353)
and the three canned enzymes (:356-362).
"""
from __future__ import annotations

import re
from dataclasses import dataclass

from . import seqhash
from .pcr import _revcomp


@dataclass
class Part:  # clone.go:57-62
    Sequence: str
    Circular: bool


@dataclass(frozen=True)
class Fragment:  # clone.go:71-76
    Sequence: str
    ForwardOverhang: str
    ReverseOverhang: str


@dataclass
class Enzyme:  # clone.go:79-86
    Name: str
    RegexpFor: "re.Pattern"
    RegexpRev: "re.Pattern"
    Skip: int
    OverheadLength: int
    RecognitionSite: str


def GetBaseRestrictionEnzymes():  # clone.go:356-362
    return [Enzyme("BsaI", re.compile("GGTCTC"), re.compile("GAGACC"), 1, 4, "GGTCTC"),
            Enzyme("BbsI", re.compile("GAAGAC"), re.compile("GTCTTC"), 2, 4, "GAAGAC"),
            Enzyme("BtgZI", re.compile("GCGATG"), re.compile("CATCGC"), 10, 4, "GCGATG")]


def _rc(s: str) -> str:
    return _revcomp(s.encode("latin-1")).decode("latin-1")


def CutWithEnzyme(part: Part, directional: bool, enzyme: Enzyme) -> list[Fragment]:
    """clone.go:135-268"""
    seq = (part.Sequence + part.Sequence if part.Circular else part.Sequence).upper()
    palindromic = enzyme.RecognitionSite == _rc(enzyme.RecognitionSite)  # checks.IsPalindromic
    span = len(enzyme.RecognitionSite) + enzyme.Skip
    fwd = [(m.end() + enzyme.Skip, True) for m in enzyme.RegexpFor.finditer(seq)]
    rev = [] if palindromic else [(m.start() - enzyme.Skip, False) for m in enzyme.RegexpRev.finditer(seq)]
    cuts = []
    for group in (fwd, rev):  # a last cut whose overhang runs off a linear sequence is dropped (:163-170)
        if group and not part.Circular and group[-1][0] + enzyme.Skip + enzyme.OverheadLength > len(seq):
            group = group[:-1]
        cuts += group
    cuts.sort(key=lambda c: c[0])  # stable, like sort.SliceStable
    oh = enzyme.OverheadLength
    if len(cuts) == 1 and not directional and not part.Circular:  # :182-203
        pos = cuts[0][0]
        if fwd:
            return [Fragment(seq[pos + oh:], seq[pos:pos + oh], ""), Fragment(seq[:pos], "", seq[pos:pos + oh])]
        return [Fragment(seq[:pos - oh], "", seq[pos - oh:pos]), Fragment(seq[pos:], seq[pos - oh:pos], "")]
    if len(cuts) == 2 and not directional and part.Circular:  # :208-216
        pos = cuts[0][0]
        return [Fragment(seq[pos + oh:len(part.Sequence)] + seq[:pos], seq[pos:pos + oh], seq[pos:pos + oh])]
    pieces = []
    for (pos, is_fwd), (npos, n_fwd) in zip(cuts, cuts[1:]):  # :227-251
        if not (directional and not palindromic) or (is_fwd and not n_fwd):
            pieces.append(seq[pos:npos])
        if npos - span > len(part.Sequence):
            break
    return [Fragment(p[oh:len(p) - oh], p[:oh], p[len(p) - oh:]) for p in pieces if len(p) > 8]  # :254-265


def _ligate(seed: Fragment, pool: list[Fragment], used: tuple, seen: set, hash_of) -> tuple[list, list]:
    """recurseLigate (clone.go:269-318) with its two seqhash.Hash calls (:275 circular, :305 linear) behind
    `hash_of(construct, circular)`.  Everything else is the reference's control flow, including what it drops:
    an endless ligation returns from the WHOLE call (:306-311), discarding what this frame had collected, while
    the hashes of the discarded constructs stay in the map."""
    if seed.ForwardOverhang == seed.ReverseOverhang:
        construct = seed.ForwardOverhang + seed.Sequence
        h = hash_of(construct, True)
        if h in seen:
            return [], []
        seen.add(h)
        return [construct], []
    out, inf = [], []
    for new in pool:
        nxt = None
        if seed.ReverseOverhang == new.ForwardOverhang:
            nxt = Fragment(seed.Sequence + seed.ReverseOverhang + new.Sequence, seed.ForwardOverhang, new.ReverseOverhang)
        if seed.ReverseOverhang == _rc(new.ReverseOverhang) and seed.ReverseOverhang != _rc(seed.ReverseOverhang):
            nxt = Fragment(seed.Sequence + seed.ReverseOverhang + _rc(new.Sequence), seed.ForwardOverhang,
                           _rc(new.ForwardOverhang))
        if nxt is None:
            continue
        for u in used:
            if u.Sequence == new.Sequence:
                construct = u.ForwardOverhang + u.Sequence + u.ReverseOverhang
                h = hash_of(construct, False)
                if h in seen:
                    return [], []
                seen.add(h)
                return [], [construct]
        used = used + (new,)  # :314 reassigns the function's own slice: it grows across siblings too
        o, i = _ligate(nxt, pool, used, seen, hash_of)
        out += o
        inf += i
    return out, inf


def dedup_by_seqhash(circular_constructs: list[str], linear_constructs: list[str]):
    """The hashing half of clone's dedup, batched: seqhash.Hash(c, "DNA", True, True) for every circular
    candidate and seqhash.Hash(c, "DNA", False, True) for every linear one -- one device call per group.
    A construct the reference could not hash maps to "" (clone.go:275 drops Hash's error)."""
    hc = seqhash.HashBatch(circular_constructs, "DNA", True, True) if circular_constructs else []
    hl = seqhash.HashBatch(linear_constructs, "DNA", False, True) if linear_constructs else []
    return ["" if isinstance(h, Exception) else h for h in hc], ["" if isinstance(h, Exception) else h for h in hl]


def CircularLigate(fragments: list[Fragment]):
    """clone.go:321-335 -> (constructs, infinite-loop constructs), deduplicated by seqhash as :275-279 / :305-309.

    Which constructs get hashed, and in which order, does not depend on any hash (both call sites are leaves), so the
    recursion runs twice: a dry pass that only RECORDS the hash calls, one batched device call per flag group for
    all of them, and the real pass fed from those results -- the reference's map logic and early returns untouched."""
    calls: list[tuple[str, bool]] = []

    def record(construct, circular):
        calls.append((construct, circular))
        return len(calls)  # a value never seen before: the dry pass takes the "new hash" branch everywhere

    for f in fragments:
        _ligate(f, fragments, (), set(), record)
    hc, hl = dedup_by_seqhash([c for c, circ in calls if circ], [c for c, circ in calls if not circ])
    ic, il = iter(hc), iter(hl)
    hashes = [next(ic) if circ else next(il) for _, circ in calls]
    at = iter(range(len(calls)))

    def replay(construct, circular):
        k = next(at)
        assert calls[k] == (construct, circular)
        return hashes[k]

    out, inf, seen = [], [], set()
    for f in fragments:
        o, i = _ligate(f, fragments, (), seen, replay)
        out += o
        inf += i
    return out, inf


def GoldenGate(sequences: list[Part], cuttingEnzyme: Enzyme):
    """clone.go:345-353"""
    fragments = []
    for part in sequences:
        fragments += CutWithEnzyme(part, True, cuttingEnzyme)
    return CircularLigate(fragments)
