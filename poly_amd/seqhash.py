"""seqhash of bebop/poly on MI355X.

Mirrors seqhash/seqhash.go: ``RotateSequence`` (:127-138, on top of
boothLeastRotation :78-124) plus the batch entry points a GPU needs.  The
rotation runs in HIP (polyhip_least_rotation_*).
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
