"""io/fastq read feeder of bebop/poly on MI355X.

Mirrors the record semantics of io/fastq/fastq.go:84-216 ((*Parser).ParseNext / ParseN): a FASTQ image goes
in, the packed (bytes, offsets) batch the hot-path kernels consume comes out -- parsed on the device
(polyhip_fastq_pack*), no per-read host objects.  Identifiers / optionals / quality strings stay in the
file image; ``rec_start`` lets the host slice them lazily.
"""
from __future__ import annotations

import numpy as np

from . import _lib

ERRORS = {
    1: "did not find fastq start '@', got to line {line}",             # fastq.go:204
    2: "empty fastq sequence, got to line {line}",                      # fastq.go:177
    3: "empty quality sequence, got to line {line}",                    # fastq.go:198
    4: "line {line} failed: unexepcted EOF encountered",                # fastq.go:147 (sic)
    5: "panic: index out of range [0] (empty identifier line {line})",  # fastq.go:156
    6: "panic: index out of range [1] (identifier field without '=' on line {line})",  # fastq.go:163
    7: "more records than max_records",
}


class FastqError(Exception):
    pass


def pack(data, max_records: int | None = None):
    """Host-pointer entry point: FASTQ bytes -> (seqs uint8, offsets uint64[n+1], rec_start uint64[n], error | None).
    Like ParseN, the records before the first bad one are returned together with the error."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, np.uint8)
    nbytes = len(buf)
    most = nbytes // 7 + 1  # shortest record: "@\nA\n\nI\n" (the third line is never looked at, fastq.go:182)
    cap = most if max_records is None else min(most, max_records)
    seqs = np.zeros(max(nbytes, 1), dtype=np.uint8)
    offsets = np.zeros(most + 2, dtype=np.uint64)
    rec = np.zeros(most + 1, dtype=np.uint64)
    result = np.zeros(4, dtype=np.uint64)
    _lib.check(_lib.lib().polyhip_fastq_pack(buf.ctypes.data if nbytes else None, nbytes, seqs.ctypes.data,
                                             offsets.ctypes.data, rec.ctypes.data, cap, result.ctypes.data))
    n, code, line, total = (int(x) for x in result)
    err = FastqError(ERRORS[code].format(line=line)) if code else None
    return seqs[:total], offsets[: n + 1], rec[:n], err


def sequences(data) -> list[bytes]:
    """The Sequence field of every record ParseAll would return (raises the parse error, if any, after them)."""
    seqs, offs, _, err = pack(data)
    out = [seqs[int(offs[i]): int(offs[i + 1])].tobytes() for i in range(len(offs) - 1)]
    if err is not None:
        raise err
    return out


def workspace_bytes(nbytes: int) -> int:
    return int(_lib.lib().polyhip_fastq_workspace_bytes(nbytes))


def pack_dev(file_t, seqs_t, offsets_t, rec_start_t, result_t, work_t, max_records: int | None = None, stream=None):
    """Device-resident feeder on torch CUDA tensors; result_t int64[4] = (n, code, line, sequence bytes)."""
    nbytes = file_t.numel()
    cap = nbytes // 7 + 1 if max_records is None else max_records
    _lib.check(_lib.lib().polyhip_fastq_pack_dev(
        file_t.data_ptr(), nbytes, seqs_t.data_ptr(), offsets_t.data_ptr(),
        rec_start_t.data_ptr() if rec_start_t is not None else None, cap, result_t.data_ptr(), work_t.data_ptr(),
        work_t.numel() * work_t.element_size(), _lib.stream_ptr(stream)))
