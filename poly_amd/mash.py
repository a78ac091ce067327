"""search/mash of bebop/poly on MI355X.

Mirrors search/mash/mash.go: ``Mash{KmerSize, SketchSize, Sketches}``, ``New``,
``(*Mash).Sketch`` (:68-104), ``Similarity`` (:107-135), ``Distance``
(:138-140), plus the batch entry points a GPU needs.
"""
from __future__ import annotations

import numpy as np

from . import _lib


def _pack(seqs):
    """list of str/bytes -> (uint8 buffer, uint64 offsets)"""
    bs = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seqs]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    buf = np.frombuffer(b"".join(bs), dtype=np.uint8) if bs else np.zeros(0, np.uint8)
    return np.ascontiguousarray(buf), offs


class Mash:
    """mash.go:52-56"""

    def __init__(self, kmer_size: int, sketch_size: int):
        self.KmerSize = int(kmer_size)
        self.SketchSize = int(sketch_size)
        self.Sketches = np.zeros(self.SketchSize, dtype=np.uint32)  # mash.go:63

    def Sketch(self, sequence) -> None:
        """mash.go:68-104: updates ``Sketches`` in place."""
        buf, offs = _pack([sequence])
        sketch_batch_packed(buf, offs, self.KmerSize, self.SketchSize,
                            out=self.Sketches.reshape(1, -1))


def New(kmer_size: int, sketch_size: int) -> Mash:
    """mash.go:59-65"""
    return Mash(kmer_size, sketch_size)


def sketch_batch_packed(seqs: np.ndarray, offsets: np.ndarray, k: int, s: int,
                        out: np.ndarray | None = None) -> np.ndarray:
    """Host-pointer entry point (what the cgo shim calls): packed batch in, (n, s) uint32 out.

    ``out`` is in/out: rows carry the prior ``Sketches`` (zeros from ``New``)."""
    n = len(offsets) - 1
    if k < 0:
        raise _lib.GoPanic(_lib.ERR_PANIC, "slice bounds out of range (negative KmerSize)")
    if s < 0:
        raise _lib.GoPanic(_lib.ERR_PANIC, "makeslice: len out of range")
    if out is None:
        out = np.zeros((n, s), dtype=np.uint32)
    assert out.dtype == np.uint32 and out.flags.c_contiguous and out.size == n * s
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    _lib.check(_lib.lib().polyhip_mash_sketch_batch(
        seqs.ctypes.data, offsets.ctypes.data, n, k, s, out.ctypes.data))
    return out


def SketchBatch(seqs, k: int, s: int) -> list[Mash]:
    """Additive batch API: one ``*Mash`` per input sequence."""
    buf, offs = _pack(seqs)
    sk = sketch_batch_packed(buf, offs, k, s)
    res = []
    for i in range(len(seqs)):
        m = Mash(k, s)
        m.Sketches = sk[i]
        res.append(m)
    return res


def sketch_batch_dev(seqs_t, offsets_t, k: int, s: int, out_t, stream=None) -> None:
    """Device-resident entry point: torch CUDA tensors (uint8 bytes, int64/uint64
    offsets, (n, s) int32/uint32 out); enqueued on ``stream`` (default: current)."""
    n = offsets_t.numel() - 1
    assert seqs_t.is_cuda and offsets_t.is_cuda and out_t.is_cuda
    assert seqs_t.is_contiguous() and offsets_t.is_contiguous() and out_t.is_contiguous()
    assert offsets_t.element_size() == 8 and out_t.element_size() == 4 and out_t.numel() == n * s
    _lib.check(_lib.lib().polyhip_mash_sketch_batch_dev(
        seqs_t.data_ptr(), offsets_t.data_ptr(), n, k, s, out_t.data_ptr(), _lib.stream_ptr(stream)))


def synth_dna_dev(seed: int, out_t, first: int = 0, stream=None) -> None:
    """Fill a CUDA uint8 tensor with the SURVEY-8d synthetic DNA stream."""
    assert out_t.is_cuda and out_t.is_contiguous() and out_t.element_size() == 1
    _lib.check(_lib.lib().polyhip_synth_dna_dev(seed & 0xFFFFFFFFFFFFFFFF, first, out_t.data_ptr(),
                                                out_t.numel(), _lib.stream_ptr(stream)))
