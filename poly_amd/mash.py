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


# ---- K2: Similarity / Distance (mash.go:107-140) ------------------------------------

def distance_matrix_packed(X: np.ndarray, Y: np.ndarray, want_counts: bool = True, want_dist: bool = True):
    """Host-pointer entry point: X (nx, sx) and Y (ny, sy) uint32 sketches ->
    (counts uint16 (nx, ny) | None, dist float64 (nx, ny) | None); receiver = X row."""
    X = np.ascontiguousarray(X, dtype=np.uint32)
    Y = np.ascontiguousarray(Y, dtype=np.uint32)
    nx, sx = X.shape
    ny, sy = Y.shape
    counts = np.zeros((nx, ny), dtype=np.uint16) if want_counts else None
    dist = np.zeros((nx, ny), dtype=np.float64) if want_dist else None
    _lib.check(_lib.lib().polyhip_mash_distance_matrix(
        X.ctypes.data, nx, sx, Y.ctypes.data, ny, sy,
        counts.ctypes.data if counts is not None else None, dist.ctypes.data if dist is not None else None))
    return counts, dist


def _similarity(self: "Mash", other: "Mash") -> float:
    """mash.go:107-135"""
    if self.SketchSize == 0 or other.SketchSize == 0:
        raise _lib.GoPanic(_lib.ERR_PANIC, "index out of range [-1] (mash.go:117)")
    counts, _ = distance_matrix_packed(self.Sketches.reshape(1, -1), other.Sketches.reshape(1, -1), True, False)
    return float(counts[0, 0]) / float(min(self.SketchSize, other.SketchSize))


def _distance(self: "Mash", other: "Mash") -> float:
    """mash.go:138-140"""
    _, dist = distance_matrix_packed(self.Sketches.reshape(1, -1), other.Sketches.reshape(1, -1), False, True)
    return float(dist[0, 0])


Mash.Similarity = _similarity
Mash.Distance = _distance


def DistanceMatrix(sketches: list[Mash]) -> np.ndarray:
    """Additive batch API: dist[i][j] = sketches[i].Distance(sketches[j]); one SketchSize."""
    S = np.stack([m.Sketches for m in sketches]).astype(np.uint32)
    return distance_matrix_packed(S, S, False, True)[1]


def sketch_distance_matrix_packed(seqs: np.ndarray, offsets: np.ndarray, k: int, s: int, want_sketches: bool = True,
                                  want_counts: bool = True, want_dist: bool = True, prior: np.ndarray | None = None):
    """Host-pointer entry point (polyhip_mash_sketch_distance_matrix): reads -> (sketches | None, counts | None,
    dist | None) without the sketches leaving HBM in between; on a device list the reads shard, the devices exchange
    their sketches by peer copies and each joins its block of rows."""
    n = len(offsets) - 1
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    sk = None
    if want_sketches or prior is not None:
        sk = np.zeros((n, s), dtype=np.uint32) if prior is None else np.ascontiguousarray(prior, dtype=np.uint32)
        assert sk.shape == (n, s)
    counts = np.zeros((n, n), dtype=np.uint16) if want_counts else None
    dist = np.zeros((n, n), dtype=np.float64) if want_dist else None
    _lib.check(_lib.lib().polyhip_mash_sketch_distance_matrix(
        seqs.ctypes.data, offsets.ctypes.data, n, k, s, sk.ctypes.data if sk is not None else None,
        counts.ctypes.data if counts is not None else None, dist.ctypes.data if dist is not None else None))
    return sk, counts, dist


def sketch_distance_matrix_last_path() -> int:
    """0 = one device, 1 = a device list with the item exchange, 2 = a device list with the sketch gather (tests)"""
    return int(_lib.lib().polyhip_mash_sketch_distance_matrix_last_path())


def sketch_distance_matrix_last_info() -> dict:
    """polyhip_matrix_info of the calling thread's last sketch_distance_matrix_packed: path, devices, device-to-device copies
    by transport (peer / staged through the host / local) with their bytes, wall ms of the sketch / index / join rounds"""
    import ctypes as C

    class Info(C.Structure):
        _fields_ = [("path", C.c_int32), ("devices", C.c_int32), ("peer_copies", C.c_int32), ("staged_copies", C.c_int32),
                    ("local_copies", C.c_int32), ("reserved", C.c_int32), ("bytes_peer", C.c_uint64), ("bytes_staged", C.c_uint64),
                    ("bytes_local", C.c_uint64), ("ms_sketch", C.c_double), ("ms_index", C.c_double), ("ms_join", C.c_double)]
    info = Info()
    _lib.check(_lib.lib().polyhip_mash_sketch_distance_matrix_last_info(C.addressof(info)))
    return {name: getattr(info, name) for name, _ in Info._fields_ if name != "reserved"}


def SketchDistanceMatrix(seqs, k: int, s: int) -> np.ndarray:
    """Additive batch API (BASELINE configs[2]): dist[i][j] = Sketch(seqs[i]).Distance(Sketch(seqs[j]))."""
    buf, offs = _pack(seqs)
    return sketch_distance_matrix_packed(buf, offs, k, s, False, False, True)[2]


def shared_counts_workspace_bytes(nx: int, sx: int, ny: int, sy: int) -> int:
    return int(_lib.lib().polyhip_mash_shared_counts_workspace_bytes(nx, sx, ny, sy))


def shared_counts_dev(X_t, Y_t, counts_t, work_t, stream=None) -> None:
    """Device-resident K2 on torch CUDA tensors: X (nx, sx) / Y (ny, sy) int32|uint32,
    counts (nx, ld >= ny) int16|uint16, work uint8[shared_counts_workspace_bytes]."""
    nx, sx = X_t.shape
    ny, sy = Y_t.shape
    assert X_t.is_cuda and Y_t.is_cuda and counts_t.is_cuda and work_t.is_cuda
    assert X_t.is_contiguous() and Y_t.is_contiguous() and X_t.element_size() == 4 and Y_t.element_size() == 4
    assert counts_t.element_size() == 2 and counts_t.shape[0] == nx and counts_t.stride(1) == 1
    ld = counts_t.stride(0)
    _lib.check(_lib.lib().polyhip_mash_shared_counts_dev(
        X_t.data_ptr(), nx, sx, Y_t.data_ptr(), ny, sy, counts_t.data_ptr(), ld,
        work_t.data_ptr(), work_t.numel() * work_t.element_size(), _lib.stream_ptr(stream)))


def index_build_dev(Y_t, work_t, stream=None) -> None:
    """Builds Y's inverted index into `work_t` (sized by shared_counts_workspace_bytes for the largest X to come)."""
    ny, sy = Y_t.shape
    assert Y_t.is_cuda and Y_t.is_contiguous() and Y_t.element_size() == 4
    _lib.check(_lib.lib().polyhip_mash_index_build_dev(Y_t.data_ptr(), ny, sy, work_t.data_ptr(),
                                                       work_t.numel() * work_t.element_size(), _lib.stream_ptr(stream)))


def index_build_part_dev(Y_t, part: int, nparts: int, work_t, stream=None) -> None:
    """One part of Y's index (polyhip_mash_index_build_part_dev): the multi-rank build, rank r builds part r of nranks."""
    ny, sy = Y_t.shape
    assert Y_t.is_cuda and Y_t.is_contiguous() and Y_t.element_size() == 4
    _lib.check(_lib.lib().polyhip_mash_index_build_part_dev(Y_t.data_ptr(), ny, sy, part, nparts, work_t.data_ptr(),
                                                            work_t.numel() * work_t.element_size(), _lib.stream_ptr(stream)))


def index_part_spans(ny: int, sy: int, nparts: int, work_t, stream=None):
    """(item_spans, start_spans): nparts + 1 byte offsets into the workspace each; part p = [spans[p], spans[p+1])."""
    it = np.zeros(nparts + 1, dtype=np.uint64)
    st = np.zeros(nparts + 1, dtype=np.uint64)
    _lib.check(_lib.lib().polyhip_mash_index_part_spans(ny, sy, nparts, work_t.data_ptr(),
                                                        work_t.numel() * work_t.element_size(), it.ctypes.data,
                                                        st.ctypes.data, _lib.stream_ptr(stream)))
    return it, st


def index_item_bytes(work_t) -> int:
    """8 (value, id | occurrence) or 4 (compact: the item carries the LDS counter it bumps) -- polyhip_mash_index_format_dev"""
    import ctypes as C
    b = C.c_uint32()
    _lib.check(_lib.lib().polyhip_mash_index_format_dev(work_t.data_ptr(), C.addressof(b)))
    return int(b.value)


def index_build_info(work_t) -> dict:
    """which index build ran and its geometry -- polyhip_mash_index_build_info_dev"""
    import ctypes as C
    info = (C.c_uint32 * 6)()
    _lib.check(_lib.lib().polyhip_mash_index_build_info_dev(work_t.data_ptr(), info))
    return dict(zip(("build", "coarse", "parts", "coarse_per_part", "two_pass_buckets", "repeated"), (int(x) for x in info)))


def index_finalize_dev(ny: int, sy: int, work_t, stream=None) -> None:
    _lib.check(_lib.lib().polyhip_mash_index_finalize_dev(ny, sy, work_t.data_ptr(),
                                                          work_t.numel() * work_t.element_size(), _lib.stream_ptr(stream)))


def shared_counts_reuse_dev(X_t, Y_t, counts_t, work_t, stream=None) -> None:
    """shared_counts_dev against the index an earlier index_build_dev / shared_counts_dev call with the same Y left in
    `work_t`: row blocks of one matrix, or queries against a resident sketch set, build the index once."""
    nx, sx = X_t.shape
    ny, sy = Y_t.shape
    assert X_t.is_cuda and Y_t.is_cuda and counts_t.is_cuda and work_t.is_cuda
    assert X_t.is_contiguous() and Y_t.is_contiguous() and X_t.element_size() == 4 and Y_t.element_size() == 4
    assert counts_t.element_size() == 2 and counts_t.shape[0] == nx and counts_t.stride(1) == 1
    _lib.check(_lib.lib().polyhip_mash_shared_counts_reuse_dev(
        X_t.data_ptr(), nx, sx, Y_t.data_ptr(), ny, sy, counts_t.data_ptr(), counts_t.stride(0),
        work_t.data_ptr(), work_t.numel() * work_t.element_size(), _lib.stream_ptr(stream)))


def shared_counts_mode(work_t):
    """(mode, irregular X, irregular Y, overflow rows, index self-join size) of the last
    shared_counts_dev on this workspace."""
    import ctypes as C
    m, ix, iy, ov, est = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
    _lib.check(_lib.lib().polyhip_mash_shared_counts_mode_dev(work_t.data_ptr(), C.addressof(m), C.addressof(ix),
                                                             C.addressof(iy), C.addressof(ov), C.addressof(est)))
    return m.value, ix.value, iy.value, ov.value, est.value


def distance_from_counts_dev(counts_t, sx: int, sy: int, dist_t, stream=None) -> None:
    nx, ny = counts_t.shape
    assert counts_t.is_cuda and dist_t.is_cuda and dist_t.element_size() == 8 and dist_t.shape == counts_t.shape
    _lib.check(_lib.lib().polyhip_mash_distance_from_counts_dev(
        counts_t.data_ptr(), nx, ny, counts_t.stride(0), sx, sy, dist_t.data_ptr(), dist_t.stride(0),
        _lib.stream_ptr(stream)))
