"""search/align of bebop/poly on MI355X.

Mirrors search/align/align.go: ``Scoring`` (:73-76), ``NewScoring`` (:79-87),
``Scoring.Score`` (:89-95), ``SmithWaterman`` (:171-232), plus the batch entry
points a GPU needs.  The DP runs in HIP (polyhip_sw_*); the substitution
matrix and alphabets stay host objects and are flattened through their public
``Score()`` exactly as the Go wrapper must (the score table is unexported).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib, alphabet, matrix
from .mash import _pack


class Scoring:
    """align.go:73-76"""

    def __init__(self, substitution_matrix: matrix.SubstitutionMatrix, gap_penalty: int):
        self.SubstitutionMatrix = substitution_matrix
        self.GapPenalty = int(gap_penalty)
        self._handle = None

    def Score(self, a: int, b: int) -> int:
        """align.go:89-95: string(byte) keys, so a byte >= 0x80 is a 2-byte string"""
        return self.SubstitutionMatrix.Score(_go_string(a), _go_string(b))

    # -- flatten + device handle -------------------------------------------
    def flatten(self):
        lut = np.zeros((256, 256), dtype=np.int32)
        va = np.zeros(256, dtype=np.uint8)
        vb = np.zeros(256, dtype=np.uint8)
        m = self.SubstitutionMatrix
        for a in range(128):
            try:
                m.FirstAlphabet.Encode(chr(a))
                va[a] = 1
            except alphabet.Error:
                pass
            try:
                m.SecondAlphabet.Encode(chr(a))
                vb[a] = 1
            except alphabet.Error:
                pass
        for a in np.nonzero(va)[0]:
            for b in np.nonzero(vb)[0]:
                lut[a, b] = m.Score(chr(a), chr(b))
        return lut, va, vb

    def handle(self):
        if self._handle is None:
            lut, va, vb = self.flatten()
            h = C.c_void_p()
            _lib.check(_lib.lib().polyhip_scoring_create(lut.ctypes.data, va.ctypes.data, vb.ctypes.data,
                                                         self.GapPenalty, C.byref(h)))
            self._handle = h
        return self._handle

    def __del__(self):
        if getattr(self, "_handle", None) is not None:
            try:
                _lib.lib().polyhip_scoring_destroy(self._handle)
            except Exception:
                pass


def _go_string(b: int) -> str:
    return chr(b)  # Go string(byte b) == the rune U+00bb; for b < 0x80 a 1-byte string


def NewScoring(substitution_matrix, gap_penalty: int) -> Scoring:
    """align.go:79-87 (nil matrix -> matrix.Default; never errors)"""
    if substitution_matrix is None:
        substitution_matrix = matrix.Default
    return Scoring(substitution_matrix, gap_penalty)


def _raise_symbol(err: int):
    raise alphabet.Error(f"Symbol {chr(err & 0xFF)} not in alphabet")


def sw_batch_packed(scoring: Scoring, A: np.ndarray, offA: np.ndarray, B: np.ndarray,
                    offB: np.ndarray | None = None):
    """Host-pointer entry point: (score int64[n], endA uint32[n], endB uint32[n], err uint32[n]).
    ``offB is None`` -> one shared B for every pair."""
    n = len(offA) - 1
    A = np.ascontiguousarray(A, dtype=np.uint8)
    B = np.ascontiguousarray(B, dtype=np.uint8)
    offA = np.ascontiguousarray(offA, dtype=np.uint64)
    score = np.zeros(n, dtype=np.int64)
    endA = np.zeros(n, dtype=np.uint32)
    endB = np.zeros(n, dtype=np.uint32)
    err = np.zeros(n, dtype=np.uint32)
    if offB is not None:
        offB = np.ascontiguousarray(offB, dtype=np.uint64)
    _lib.check(_lib.lib().polyhip_sw_batch(
        scoring.handle(), A.ctypes.data, offA.ctypes.data, n, B.ctypes.data,
        offB.ctypes.data if offB is not None else None, len(B) if offB is None else 0,
        score.ctypes.data, endA.ctypes.data, endB.ctypes.data, err.ctypes.data))
    return score, endA, endB, err


def SmithWatermanScoreBatch(reads, ref, scoring: Scoring):
    """Additive batch API: every read against one shared reference."""
    A, offA = _pack(reads)
    B, _ = _pack([ref])
    return sw_batch_packed(scoring, A, offA, B, None)


def sw_workspace_bytes(scoring: Scoring, npairs: int, max_lenA: int, lenB: int, shared: bool = True) -> int:
    return int(_lib.lib().polyhip_sw_workspace_bytes(scoring.handle(), npairs, max_lenA, lenB, int(shared)))


def sw_batch_dev(scoring: Scoring, A_t, offA_t, max_lenA: int, B_t, offB_t, lenB: int,
                 score_t, endA_t, endB_t, err_t, work_t, stream=None) -> None:
    """Device-resident entry point on torch CUDA tensors."""
    n = offA_t.numel() - 1
    _lib.check(_lib.lib().polyhip_sw_batch_dev(
        scoring.handle(), A_t.data_ptr(), offA_t.data_ptr(), n, max_lenA, B_t.data_ptr(),
        offB_t.data_ptr() if offB_t is not None else None, lenB,
        score_t.data_ptr(), endA_t.data_ptr(), endB_t.data_ptr(), err_t.data_ptr(),
        work_t.data_ptr(), work_t.numel() * work_t.element_size(), _lib.stream_ptr(stream)))


def last_path() -> int:
    return int(_lib.lib().polyhip_sw_last_path())


def last_packed_half() -> bool:
    """the last packed pass ran gfx950's half-float cell (scores below 2048)"""
    return bool(_lib.lib().polyhip_sw_last_packed_half())


def last_packed_lanes() -> int:
    """lanes that shared the rows of a lane's two read pairs in the last packed score pass (0: none ran)"""
    return int(_lib.lib().polyhip_sw_last_packed_lanes())


def sw_traceback_last_path() -> int:
    """1 = byte-profile traceback kernel, 2 = register-tiled table kernel, 3 = generic (tests)"""
    return int(_lib.lib().polyhip_sw_traceback_last_path())


def sw_traceback_last_half() -> bool:
    """the last byte-profile traceback ran gfx950's half-float two-band form"""
    return bool(_lib.lib().polyhip_sw_traceback_last_half())


def nw_last_path() -> int:
    """1 = register-tiled NeedlemanWunsch kernel, 2 = generic (tests)"""
    return int(_lib.lib().polyhip_nw_last_path())


# ---- full SmithWaterman: score pass + traceback (align.go:171-232) -------------------

def sw_align_packed(scoring: Scoring, A: np.ndarray, offA: np.ndarray, B: np.ndarray,
                    offB: np.ndarray | None = None):
    """Host-pointer entry point: (score, endA, endB, err, alignA list[bytes], alignB list[bytes])."""
    n = len(offA) - 1
    A = np.ascontiguousarray(A, dtype=np.uint8)
    B = np.ascontiguousarray(B, dtype=np.uint8)
    offA = np.ascontiguousarray(offA, dtype=np.uint64)
    if offB is not None:
        offB = np.ascontiguousarray(offB, dtype=np.uint64)
    lensA = np.diff(offA.astype(np.int64))
    max_lenA = int(lensA.max()) if n else 0
    lenB = len(B) if offB is None else (int(np.diff(offB.astype(np.int64)).max()) if n else 0)
    stride = int(_lib.lib().polyhip_sw_traceback_stride(scoring.handle(), max_lenA, lenB))
    score = np.zeros(n, dtype=np.int64)
    endA = np.zeros(n, dtype=np.uint32)
    endB = np.zeros(n, dtype=np.uint32)
    err = np.zeros(n, dtype=np.uint32)
    alnA = np.zeros((n, stride), dtype=np.uint8)
    alnB = np.zeros((n, stride), dtype=np.uint8)
    alen = np.zeros(n, dtype=np.uint32)
    _lib.check(_lib.lib().polyhip_sw_align_batch(
        scoring.handle(), A.ctypes.data, offA.ctypes.data, n, B.ctypes.data,
        offB.ctypes.data if offB is not None else None, len(B) if offB is None else 0,
        score.ctypes.data, endA.ctypes.data, endB.ctypes.data, err.ctypes.data,
        alnA.ctypes.data, alnB.ctypes.data, alen.ctypes.data, stride))
    sa = [alnA[p, stride - int(alen[p]):].tobytes() for p in range(n)]
    sb = [alnB[p, stride - int(alen[p]):].tobytes() for p in range(n)]
    return score, endA, endB, err, sa, sb


def sw_align_strings_packed(scoring: Scoring, A: np.ndarray, offA: np.ndarray, B: np.ndarray, offB: np.ndarray | None = None,
                            capacity: int | None = None):
    """Host-pointer entry point with PACKED strings (polyhip_sw_align_batch_packed: only the strings' own bytes cross
    PCIe): (score, endA, endB, err, alignA list[bytes], alignB list[bytes]).  `capacity` bytes per string buffer
    (default: 1.25 x the reads' bytes + 64 KB; a batch that needs more is run again with the exact size)."""
    n = len(offA) - 1
    A = np.ascontiguousarray(A, dtype=np.uint8)
    B = np.ascontiguousarray(B, dtype=np.uint8)
    offA = np.ascontiguousarray(offA, dtype=np.uint64)
    if offB is not None:
        offB = np.ascontiguousarray(offB, dtype=np.uint64)
    score = np.zeros(n, dtype=np.int64)
    endA, endB, err = (np.zeros(n, dtype=np.uint32) for _ in range(3))
    off = np.zeros(n + 1, dtype=np.uint64)
    cap = int(capacity) if capacity is not None else int(int(offA[n] - offA[0]) * 1.25) + (64 << 10)
    for _ in range(2):
        alnA, alnB = np.zeros(max(cap, 1), dtype=np.uint8), np.zeros(max(cap, 1), dtype=np.uint8)
        rc = _lib.lib().polyhip_sw_align_batch_packed(
            scoring.handle(), A.ctypes.data, offA.ctypes.data, n, B.ctypes.data,
            offB.ctypes.data if offB is not None else None, len(B) if offB is None else 0,
            score.ctypes.data, endA.ctypes.data, endB.ctypes.data, err.ctypes.data, alnA.ctypes.data, alnB.ctypes.data,
            off.ctypes.data, cap)
        if rc == _lib.ERR_INVALID and int(off[n]) > cap:   # the strings did not fit: off[n] says what they need
            cap = int(off[n])
            continue
        _lib.check(rc)
        break
    o = off.astype(np.int64)
    sa = [alnA[o[p]:o[p + 1]].tobytes() for p in range(n)]
    sb = [alnB[o[p]:o[p + 1]].tobytes() for p in range(n)]
    return score, endA, endB, err, sa, sb


def SmithWaterman(stringA, stringB, scoring: Scoring):
    """align.go:171-232 -> (score, alignA, alignB); raises alphabet.Error like the reference's err."""
    A, offA = _pack([stringA])
    B, _ = _pack([stringB])
    score, _, _, err, sa, sb = sw_align_packed(scoring, A, offA, B, None)
    if err[0]:
        _raise_symbol(int(err[0]))
    return int(score[0]), sa[0].decode("latin-1"), sb[0].decode("latin-1")


def SmithWatermanBatch(reads, ref, scoring: Scoring):
    """Additive batch API: every read against one shared reference ->
    list of (score, alignA, alignB) or alphabet.Error instances."""
    A, offA = _pack(reads)
    B, _ = _pack([ref])
    score, _, _, err, sa, sb = sw_align_strings_packed(scoring, A, offA, B, None)
    out = []
    for p in range(len(reads)):
        if err[p]:
            out.append(alphabet.Error(f"Symbol {chr(int(err[p]) & 0xFF)} not in alphabet"))
        else:
            out.append((int(score[p]), sa[p].decode("latin-1"), sb[p].decode("latin-1")))
    return out


def sw_traceback_dev(scoring: Scoring, A_t, offA_t, max_lenA: int, B_t, offB_t, lenB: int, endA_t, endB_t, err_t,
                     alnA_t, alnB_t, alnLen_t, work_t, stream=None, score_t=None) -> None:
    """Device-resident traceback on torch CUDA tensors (alnA/alnB: (n, stride) uint8); score_t (the
    score pass's int64 output) lets every pair shrink its window."""
    n = offA_t.numel() - 1
    _lib.check(_lib.lib().polyhip_sw_traceback_dev(
        scoring.handle(), A_t.data_ptr(), offA_t.data_ptr(), n, max_lenA, B_t.data_ptr(),
        offB_t.data_ptr() if offB_t is not None else None, lenB, endA_t.data_ptr(), endB_t.data_ptr(),
        err_t.data_ptr(), score_t.data_ptr() if score_t is not None else None, alnA_t.data_ptr(), alnB_t.data_ptr(),
        alnLen_t.data_ptr(), alnA_t.shape[1],
        work_t.data_ptr(), work_t.numel() * work_t.element_size(), _lib.stream_ptr(stream)))


def sw_align_dev(scoring: Scoring, A_t, offA_t, max_lenA: int, B_t, offB_t, lenB: int, score_t, endA_t, endB_t, err_t,
                 alnA_t, alnB_t, alnLen_t, work_t, tb_work_t, stream=None) -> None:
    """Device-resident whole SmithWaterman (score pass + traceback in one call; the packed score pass leaves the end
    cell to the traceback kernel where it can): same outputs as sw_batch_dev + sw_traceback_dev."""
    n = offA_t.numel() - 1
    _lib.check(_lib.lib().polyhip_sw_align_batch_dev(
        scoring.handle(), A_t.data_ptr(), offA_t.data_ptr(), n, max_lenA, B_t.data_ptr(),
        offB_t.data_ptr() if offB_t is not None else None, lenB, score_t.data_ptr(), endA_t.data_ptr(), endB_t.data_ptr(),
        err_t.data_ptr(), alnA_t.data_ptr(), alnB_t.data_ptr(), alnLen_t.data_ptr(), alnA_t.shape[1],
        work_t.data_ptr(), work_t.numel() * work_t.element_size(), tb_work_t.data_ptr(),
        tb_work_t.numel() * tb_work_t.element_size(), _lib.stream_ptr(stream)))


def sw_traceback_stride(scoring: Scoring, max_lenA: int, lenB: int) -> int:
    return int(_lib.lib().polyhip_sw_traceback_stride(scoring.handle(), max_lenA, lenB))


def sw_traceback_workspace_bytes(scoring: Scoring, npairs: int, max_lenA: int, lenB: int) -> int:
    return int(_lib.lib().polyhip_sw_traceback_workspace_bytes(scoring.handle(), npairs, max_lenA, lenB))


# ---- NeedlemanWunsch (align.go:100-166) -----------------------------------------------------

def nw_align_packed(scoring: Scoring, A: np.ndarray, offA: np.ndarray, B: np.ndarray, offB: np.ndarray | None = None):
    """Host-pointer entry point: (score int64[n], err uint32[n], alignA list[bytes], alignB list[bytes])."""
    n = len(offA) - 1
    A = np.ascontiguousarray(A, dtype=np.uint8)
    B = np.ascontiguousarray(B, dtype=np.uint8)
    offA = np.ascontiguousarray(offA, dtype=np.uint64)
    if offB is not None:
        offB = np.ascontiguousarray(offB, dtype=np.uint64)
    max_lenA = int(np.diff(offA.astype(np.int64)).max()) if n else 0
    lenB = len(B) if offB is None else (int(np.diff(offB.astype(np.int64)).max()) if n else 0)
    stride = max(1, max_lenA + lenB)
    score = np.zeros(n, dtype=np.int64)
    err = np.zeros(n, dtype=np.uint32)
    alnA = np.zeros((n, stride), dtype=np.uint8)
    alnB = np.zeros((n, stride), dtype=np.uint8)
    alen = np.zeros(n, dtype=np.uint32)
    _lib.check(_lib.lib().polyhip_nw_align_batch(
        scoring.handle(), A.ctypes.data, offA.ctypes.data, n, B.ctypes.data,
        offB.ctypes.data if offB is not None else None, len(B) if offB is None else 0,
        score.ctypes.data, err.ctypes.data, alnA.ctypes.data, alnB.ctypes.data, alen.ctypes.data, stride))
    sa = [alnA[p, stride - int(alen[p]):].tobytes() for p in range(n)]
    sb = [alnB[p, stride - int(alen[p]):].tobytes() for p in range(n)]
    return score, err, sa, sb


def nw_workspace_bytes(npairs: int, max_lenA: int, max_lenB: int) -> int:
    return int(_lib.lib().polyhip_nw_workspace_bytes(npairs, max_lenA, max_lenB))


def nw_align_dev(scoring: Scoring, A_t, offA_t, max_lenA: int, B_t, offB_t, lenB: int, score_t, err_t, alnA_t, alnB_t,
                 alnLen_t, work_t, stream=None) -> None:
    """Device-resident NeedlemanWunsch on torch CUDA tensors (alnA/alnB: (n, stride >= max_lenA + lenB) uint8,
    strings right-aligned in their slots); lenB = the longest B (or the shared B's length when offB_t is None)."""
    n = offA_t.numel() - 1
    _lib.check(_lib.lib().polyhip_nw_align_batch_dev(
        scoring.handle(), A_t.data_ptr(), offA_t.data_ptr(), n, max_lenA, B_t.data_ptr(),
        offB_t.data_ptr() if offB_t is not None else None, lenB, score_t.data_ptr(), err_t.data_ptr(),
        alnA_t.data_ptr(), alnB_t.data_ptr(), alnLen_t.data_ptr(), alnA_t.shape[1], work_t.data_ptr(),
        work_t.numel() * work_t.element_size(), _lib.stream_ptr(stream)))


def NeedlemanWunsch(stringA, stringB, scoring: Scoring):
    """align.go:100-166 -> (score, alignA, alignB); raises alphabet.Error like the reference's err."""
    A, offA = _pack([stringA])
    B, _ = _pack([stringB])
    score, err, sa, sb = nw_align_packed(scoring, A, offA, B, None)
    if err[0]:
        _raise_symbol(int(err[0]))
    return int(score[0]), sa[0].decode("latin-1"), sb[0].decode("latin-1")
