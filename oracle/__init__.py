"""CPU oracle for the poly search hot path -- TEST INFRASTRUCTURE ONLY.

ctypes binding over ``oracle/libpolyoracle.so`` (built by ``make -C oracle`` or
``__graft_entry__.build()``).  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this package, and only as
the checker.  ``poly_amd`` never imports it.

Each wrapper names the reference function it restates (paths relative to
/root/reference); the restatement itself is ``poly_oracle.c``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpolyoracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "poly_oracle.c")
    hdr = os.path.join(_HERE, "poly_oracle.h")
    stale = (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    )
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libpolyoracle.so"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u8p, u32p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
        L.orc_synth_dna.argtypes = [C.c_uint64, C.c_void_p, C.c_size_t]
        L.orc_synth_dna.restype = None
        L.orc_murmur3_32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        L.orc_murmur3_32.restype = C.c_uint32
        L.orc_mash_sketch.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_mash_sketch.restype = C.c_int
        L.orc_mash_sketch_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_mash_sketch_batch.restype = C.c_int
        for f in (L.orc_mash_similarity, L.orc_mash_distance):
            f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
            f.restype = C.c_double
        L.orc_mash_shared.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_mash_shared.restype = C.c_int
        L.orc_submat_score.argtypes = [C.c_void_p, C.c_uint8, C.c_uint8, C.POINTER(C.c_int)]
        L.orc_submat_score.restype = C.c_int
        L.orc_submat_flatten.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_submat_flatten.restype = None
        L.orc_smith_waterman.argtypes = [
            C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int,
            C.POINTER(C.c_int64), C.c_char_p, C.c_char_p, C.POINTER(C.c_uint32),
            C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]
        L.orc_smith_waterman.restype = C.c_int
        L.orc_needleman_wunsch.argtypes = [
            C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int,
            C.POINTER(C.c_int64), C.c_char_p, C.c_char_p, C.POINTER(C.c_uint8)]
        L.orc_needleman_wunsch.restype = C.c_int
        L.orc_reverse_complement.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_reverse_complement.restype = None
        L.orc_go_log.argtypes = [C.c_double]
        L.orc_go_log.restype = C.c_double
        L.orc_santalucia.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_double,
                                     C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_santalucia.restype = None
        L.orc_santalucia_scan.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_double,
                                          C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_santalucia_scan.restype = None
        L.orc_mash_distance_matrix.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.orc_mash_distance_matrix.restype = None
        L.orc_marmur_doty.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_marmur_doty.restype = C.c_double
        L.orc_melting_temp.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_melting_temp.restype = C.c_double
        L.orc_booth_least_rotation.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_booth_least_rotation.restype = C.c_size_t
        L.orc_rotate_sequence.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_rotate_sequence.restype = None
        L.orc_blake3_256.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_blake3_256.restype = None
        L.orc_seqhash.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.c_char_p,
                                  C.POINTER(C.c_uint8)]
        L.orc_seqhash.restype = C.c_int
        del u8p, u32p, i32p
        _lib = L
    return _lib


def _b(x) -> bytes:
    if isinstance(x, str):
        return x.encode("latin-1")
    if isinstance(x, np.ndarray):
        return x.tobytes()
    return bytes(x)


class GoPanic(Exception):
    """The reference would panic (index out of range) on these arguments."""


# --------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------
def synth_dna(seed: int, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    lib().orc_synth_dna(seed & 0xFFFFFFFFFFFFFFFF, out.ctypes.data, n)
    return out


# --------------------------------------------------------------------------
# search/mash
# --------------------------------------------------------------------------
def murmur3_32(data, seed: int = 0) -> int:
    d = _b(data)
    return int(lib().orc_murmur3_32(d, len(d), seed))


class Mash:
    """search/mash/mash.go:52-65 (Mash, New)."""

    def __init__(self, kmer_size: int, sketch_size: int):
        self.KmerSize = kmer_size
        self.SketchSize = sketch_size
        self.Sketches = np.zeros(sketch_size, dtype=np.uint32)

    def Sketch(self, sequence, faithful: bool = False) -> None:
        """mash.go:68-104"""
        d = _b(sequence)
        rc = lib().orc_mash_sketch(d, len(d), self.KmerSize, self.SketchSize,
                                   self.Sketches.ctypes.data, int(faithful))
        if rc != 0:
            raise GoPanic("index out of range [-1]")

    def Similarity(self, other: "Mash") -> float:
        """mash.go:107-135"""
        return float(lib().orc_mash_similarity(self.Sketches.ctypes.data, self.SketchSize,
                                               other.Sketches.ctypes.data, other.SketchSize))

    def Distance(self, other: "Mash") -> float:
        """mash.go:138-140"""
        return float(lib().orc_mash_distance(self.Sketches.ctypes.data, self.SketchSize,
                                             other.Sketches.ctypes.data, other.SketchSize))


def mash_sketch_batch(seqs: np.ndarray, offsets: np.ndarray, k: int, s: int,
                      out: np.ndarray | None = None, faithful: bool = False) -> np.ndarray:
    """Loop of (*Mash).Sketch over a packed batch; ``out`` carries prior state."""
    n = len(offsets) - 1
    if out is None:
        out = np.zeros((n, s), dtype=np.uint32)
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    assert out.dtype == np.uint32 and out.flags.c_contiguous
    # one C call for the whole batch (ctypes releases the GIL: threads over shards run in parallel)
    if lib().orc_mash_sketch_batch(seqs.ctypes.data, offsets.ctypes.data, n, k, s, out.ctypes.data, int(faithful)) != 0:
        raise GoPanic("index out of range [-1]")
    return out


def mash_shared(a: np.ndarray, b: np.ndarray) -> int:
    return int(lib().orc_mash_shared(a.ctypes.data, len(a), b.ctypes.data, len(b)))


# --------------------------------------------------------------------------
# search/align
# --------------------------------------------------------------------------
class _CSubmat(C.Structure):
    _fields_ = [("na", C.c_int), ("nb", C.c_int), ("symA", C.c_char_p),
                ("symB", C.c_char_p), ("scores", C.POINTER(C.c_int))]


class AlphabetError(Exception):
    """alphabet/alphabet.go:14-22,38 -- 'Symbol X not in alphabet'."""


class SubstitutionMatrix:
    """search/align/matrix/matrix.go:13-38 over single-byte symbols."""

    def __init__(self, first: str, second: str, scores):
        sc = np.ascontiguousarray(np.array(scores, dtype=np.intc))
        if sc.shape != (len(first), len(second)):
            raise ValueError("invalid dimensions of substitution matrix")  # matrix.go:21-23
        self.first, self.second, self.scores = first, second, sc
        self._symA, self._symB = first.encode(), second.encode()
        self._c = _CSubmat(len(first), len(second), self._symA, self._symB,
                           sc.ctypes.data_as(C.POINTER(C.c_int)))

    def ptr(self):
        return C.addressof(self._c)

    def Score(self, a: str, b: str) -> int:
        v = C.c_int(0)
        e = lib().orc_submat_score(self.ptr(), ord(a), ord(b), C.byref(v))
        if e:
            raise AlphabetError(f"Symbol {a if e == 1 else b} not in alphabet")
        return v.value

    def flatten(self):
        lut = np.zeros((256, 256), dtype=np.int32)
        va = np.zeros(256, dtype=np.uint8)
        vb = np.zeros(256, dtype=np.uint8)
        lib().orc_submat_flatten(self.ptr(), lut.ctypes.data, va.ctypes.data, vb.ctypes.data)
        return lut, va, vb


_LETTERS = "ABCDEFGHIJKLMNOPQRSTUVWXYZ"
#: matrix.Default, matrix.go:40-73 (26 letters, +1 diagonal / -1 elsewhere)
DEFAULT_MATRIX = SubstitutionMatrix(_LETTERS, _LETTERS,
                                    2 * np.eye(26, dtype=np.intc) - 1)
#: matrix.NUC_4, matrices.go:33-40 with its intended alphabet order "-ACGT"
NUC_4_SCORES = [[0, 0, 0, 0, 0], [0, 5, -4, -4, -4], [0, -4, 5, -4, -4],
                [0, -4, -4, 5, -4], [0, -4, -4, -4, 5]]


def smith_waterman(a, b, mat: SubstitutionMatrix, gap: int):
    """align.go:171-232 -> (score, alignA, alignB, endA, endB); raises AlphabetError."""
    da, db = _b(a), _b(b)
    score = C.c_int64(0)
    bufA = C.create_string_buffer(len(da) + len(db) + 1)
    bufB = C.create_string_buffer(len(da) + len(db) + 1)
    ea, eb, sym = C.c_uint32(0), C.c_uint32(0), C.c_uint8(0)
    e = lib().orc_smith_waterman(da, len(da), db, len(db), mat.ptr(), gap, C.byref(score),
                                 bufA, bufB, C.byref(ea), C.byref(eb), C.byref(sym))
    if e:
        raise AlphabetError(f"Symbol {chr(sym.value)} not in alphabet")
    return score.value, bufA.value.decode("latin-1"), bufB.value.decode("latin-1"), ea.value, eb.value


def needleman_wunsch(a, b, mat: SubstitutionMatrix, gap: int):
    """align.go:100-166 -> (score, alignA, alignB)."""
    da, db = _b(a), _b(b)
    score = C.c_int64(0)
    bufA = C.create_string_buffer(len(da) + len(db) + 1)
    bufB = C.create_string_buffer(len(da) + len(db) + 1)
    sym = C.c_uint8(0)
    e = lib().orc_needleman_wunsch(da, len(da), db, len(db), mat.ptr(), gap, C.byref(score),
                                   bufA, bufB, C.byref(sym))
    if e:
        raise AlphabetError(f"Symbol {chr(sym.value)} not in alphabet")
    return score.value, bufA.value.decode("latin-1"), bufB.value.decode("latin-1")


# --------------------------------------------------------------------------
# transform / primers
# --------------------------------------------------------------------------
def reverse_complement(seq) -> bytes:
    d = _b(seq)
    out = C.create_string_buffer(len(d) + 1)
    lib().orc_reverse_complement(d, len(d), out)
    return out.raw[: len(d)]


def go_log(x: float) -> float:
    return float(lib().orc_go_log(x))


def santalucia(seq, primer_conc: float, salt_conc: float, mg_conc: float):
    """primers.go:70-105 -> (Tm, dH, dS)"""
    d = _b(seq)
    if len(d) == 0:
        raise GoPanic("index out of range [-1]")  # primers.go:89 on ""
    tm, dh, ds = C.c_double(), C.c_double(), C.c_double()
    lib().orc_santalucia(d, len(d), primer_conc, salt_conc, mg_conc,
                         C.byref(tm), C.byref(dh), C.byref(ds))
    return tm.value, dh.value, ds.value


def santalucia_scan(genome, Lmin: int, Lmax: int, primer_conc: float, salt_conc: float, mg_conc: float):
    """SantaLucia of every genome[i:i+L], L = Lmin..Lmax -> three (Lmax-Lmin+1, n-Lmin+1) planes (NaN-padded)"""
    g = np.frombuffer(_b(genome), dtype=np.uint8)
    shape = (Lmax - Lmin + 1, max(0, len(g) - Lmin + 1))
    tm, dh, ds = (np.full(shape, np.nan) for _ in range(3))
    lib().orc_santalucia_scan(g.ctypes.data, len(g), Lmin, Lmax, primer_conc, salt_conc, mg_conc,
                              tm.ctypes.data, dh.ctypes.data, ds.ctypes.data)
    return tm, dh, ds


def mash_distance_matrix(X: np.ndarray, Y: np.ndarray) -> np.ndarray:
    """(*Mash).Distance (mash.go:138-140) for every ordered pair of the rows of X and Y (sorted sketches)"""
    X = np.ascontiguousarray(X, dtype=np.uint32)
    Y = np.ascontiguousarray(Y, dtype=np.uint32)
    assert X.shape[1] == Y.shape[1]
    out = np.empty((X.shape[0], Y.shape[0]), dtype=np.float64)
    lib().orc_mash_distance_matrix(X.ctypes.data, X.shape[0], Y.ctypes.data, Y.shape[0], X.shape[1], out.ctypes.data)
    return out


def marmur_doty(seq) -> float:
    d = _b(seq)
    return float(lib().orc_marmur_doty(d, len(d)))


def melting_temp(seq) -> float:
    d = _b(seq)
    if len(d) == 0:
        raise GoPanic("index out of range [-1]")
    return float(lib().orc_melting_temp(d, len(d)))


# --------------------------------------------------------------------------
# seqhash
# --------------------------------------------------------------------------
def booth_least_rotation(seq) -> int:
    d = _b(seq)
    return int(lib().orc_booth_least_rotation(d, len(d)))


def rotate_sequence(seq) -> bytes:
    d = _b(seq)
    out = C.create_string_buffer(len(d) + 1)
    lib().orc_rotate_sequence(d, len(d), out)
    return out.raw[: len(d)]


def blake3_256(data) -> bytes:
    d = _b(data)
    out = C.create_string_buffer(32)
    lib().orc_blake3_256(d, len(d), out)
    return out.raw


class SeqhashError(Exception):
    pass


def seqhash(seq, seq_type: str, circular: bool, double_stranded: bool) -> str:
    """seqhash.go:141-224"""
    d = _b(seq)
    out = C.create_string_buffer(72)
    ch = C.c_uint8(0)
    e = lib().orc_seqhash(d, len(d), seq_type.encode(), int(circular), int(double_stranded),
                          out, C.byref(ch))
    if e == 1:
        raise SeqhashError("Only sequenceTypes of DNA, RNA, or PROTEIN allowed. Got sequenceType: " + seq_type)
    if e == 2:
        raise SeqhashError("Only letters ATUGCYRSWKMBDHVNZ are allowed for DNA/RNA. Got letter: " + chr(ch.value))
    if e == 3:
        raise SeqhashError("Only letters ACDEFGHIKLMNPQRSTVWYUO*BXZ are allowed for Proteins. Got letter: " + chr(ch.value))
    if e == 4:
        raise SeqhashError("Proteins cannot be double stranded")
    return out.value.decode()
