"""seqhash of bebop/poly on MI355X.

Mirrors seqhash/seqhash.go: ``RotateSequence`` (:127-138, on top of
boothLeastRotation :78-124) and ``Hash`` (:141-224), plus the batch entry points
a GPU needs.  Rotation, reverse complement, candidate choice and BLAKE3 all run
in HIP (polyhip_least_rotation_*, polyhip_seqhash_*).
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .mash import _pack


def least_rotation_batch_packed(seqs: np.ndarray, offsets: np.ndarray, want_rotated: bool = True):
    """Host-pointer entry point: (rot_index uint64[n], rotated uint8 buffer | None)."""
    n = len(offsets) - 1
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    rot = np.zeros(n, dtype=np.uint64)
    out = np.zeros(max(1, len(seqs)), dtype=np.uint8) if want_rotated else None
    _lib.check(_lib.lib().polyhip_least_rotation_batch(seqs.ctypes.data, offsets.ctypes.data, n, rot.ctypes.data,
                                                       out.ctypes.data if out is not None else None))
    return rot, out


def RotateSequence(sequence: str) -> str:
    """seqhash.go:127-138"""
    buf, offs = _pack([sequence])
    _, out = least_rotation_batch_packed(buf, offs, True)
    return out[: len(buf)].tobytes().decode("latin-1")


def RotateBatch(seqs) -> list[str]:
    """Additive batch API: RotateSequence of every sequence."""
    buf, offs = _pack(seqs)
    _, out = least_rotation_batch_packed(buf, offs, True)
    return [out[int(offs[i]): int(offs[i + 1])].tobytes().decode("latin-1") for i in range(len(seqs))]


def least_rotation_batch_dev(seqs_t, offsets_t, max_len: int, rot_t, rotated_t=None, stream=None) -> None:
    """Device-resident entry point on torch CUDA tensors."""
    n = offsets_t.numel() - 1
    assert seqs_t.is_cuda and offsets_t.is_cuda and rot_t.is_cuda and rot_t.element_size() == 8 and rot_t.numel() >= n
    _lib.check(_lib.lib().polyhip_least_rotation_batch_dev(
        seqs_t.data_ptr(), offsets_t.data_ptr(), n, max_len, rot_t.data_ptr(),
        rotated_t.data_ptr() if rotated_t is not None else None, _lib.stream_ptr(stream)))


# ---- Hash (seqhash.go:141-224) -----------------------------------------------------------
DNA, RNA, PROTEIN = "DNA", "RNA", "PROTEIN"  # seqhash.go:70-74
_TYPE_CODE = {DNA: 0, RNA: 1, PROTEIN: 2}


def seqhash_batch_packed(seqs: np.ndarray, offsets: np.ndarray, seq_type: int, circular: bool, double_stranded: bool):
    """Host-pointer entry point: (list of 71-char hashes ('' on error), err uint32[n])."""
    n = len(offsets) - 1
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    out = np.zeros((n, 72), dtype=np.uint8)
    err = np.zeros(n, dtype=np.uint32)
    _lib.check(_lib.lib().polyhip_seqhash_batch(seqs.ctypes.data, offsets.ctypes.data, n, seq_type, int(circular),
                                                int(double_stranded), out.ctypes.data, err.ctypes.data))
    return [out[i].tobytes().split(b"\0", 1)[0].decode("ascii") for i in range(n)], err


def _error_for(code: int) -> ValueError:
    letter = chr(code & 0xFF)
    if (code >> 8) == 2:  # seqhash.go:157
        return ValueError("Only letters ATUGCYRSWKMBDHVNZ are allowed for DNA/RNA. Got letter: " + letter)
    return ValueError("Only letters ACDEFGHIKLMNPQRSTVWYUO*BXZ are allowed for Proteins. Got letter: " + letter)  # :169


def HashBatch(sequences, sequenceType: str, circular: bool, doubleStranded: bool):
    """Additive batch API: Hash of every sequence under one (type, circular, doubleStranded);
    entries are the 71-character seqhash or a ValueError with the reference's message."""
    if sequenceType not in _TYPE_CODE:  # seqhash.go:152
        raise ValueError("Only sequenceTypes of DNA, RNA, or PROTEIN allowed. Got sequenceType: " + str(sequenceType))
    if sequenceType == PROTEIN and doubleStranded:  # seqhash.go:175 (checked after the letters in the reference)
        buf, offs = _pack(sequences)
        _, err = seqhash_batch_packed(buf, offs, 2, circular, False)
        return [_error_for(int(e)) if e else ValueError("Proteins cannot be double stranded") for e in err]
    buf, offs = _pack(sequences)
    hashes, err = seqhash_batch_packed(buf, offs, _TYPE_CODE[sequenceType], circular, doubleStranded)
    return [_error_for(int(e)) if e else h for h, e in zip(hashes, err)]


def Hash(sequence, sequenceType: str, circular: bool, doubleStranded: bool) -> str:
    """seqhash.go:141-224; raises ValueError with the reference's error text."""
    r = HashBatch([sequence], sequenceType, circular, doubleStranded)[0]
    if isinstance(r, Exception):
        raise r
    return r


def seqhash_batch_dev(seqs_t, offsets_t, total_bytes: int, max_len: int, seq_type: int, circular: bool,
                      double_stranded: bool, out_t, err_t, work_t, stream=None) -> None:
    """Device-resident entry point on torch CUDA tensors (out: (n, 72) uint8, err: int32[n])."""
    n = offsets_t.numel() - 1
    _lib.check(_lib.lib().polyhip_seqhash_batch_dev(
        seqs_t.data_ptr(), offsets_t.data_ptr(), n, total_bytes, max_len, seq_type, int(circular), int(double_stranded),
        out_t.data_ptr(), err_t.data_ptr(), work_t.data_ptr(), work_t.numel() * work_t.element_size(),
        _lib.stream_ptr(stream)))


def seqhash_workspace_bytes(n: int, total_bytes: int, circular: bool, double_stranded: bool) -> int:
    return int(_lib.lib().polyhip_seqhash_workspace_bytes(n, total_bytes, int(circular), int(double_stranded)))
