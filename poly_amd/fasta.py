"""io/fasta read feeder of bebop/poly on MI355X.

Mirrors the record semantics of io/fasta/fasta.go:102-238 ((*Parser).ParseNext / ParseN): a FASTA image goes
in, the packed (bytes, offsets) batch the hot-path kernels consume comes out -- parsed on the device
(polyhip_fasta_pack*).  Names stay in the file image; ``rec_start`` points at each record's header line.
"""
from __future__ import annotations

import numpy as np

from . import _lib

ERRORS = {
    1: "did not find fasta start '>'",   # fasta.go:223
    2: "empty fasta sequence",           # fasta.go:227
    7: "more records than max_records",
}


class FastaError(Exception):
    pass


def pack(data, max_records: int | None = None):
    """Host-pointer entry point: FASTA bytes -> (seqs uint8, offsets uint64[n+1], rec_start uint64[n], error | None).
    Like ParseN, the records before the first bad one are returned together with the error."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, np.uint8)
    nbytes = len(buf)
    most = nbytes // 2 + 2
    cap = most if max_records is None else min(most, max_records)
    seqs = np.zeros(max(nbytes, 1), dtype=np.uint8)
    offsets = np.zeros(most + 2, dtype=np.uint64)
    rec = np.zeros(most + 2, dtype=np.uint64)
    result = np.zeros(4, dtype=np.uint64)
    _lib.check(_lib.lib().polyhip_fasta_pack(buf.ctypes.data if nbytes else None, nbytes, seqs.ctypes.data,
                                             offsets.ctypes.data, rec.ctypes.data, cap, result.ctypes.data))
    n, code, total, _ = (int(x) for x in result)
    err = FastaError(ERRORS[code]) if code else None
    return seqs[:total], offsets[: n + 1], rec[:n], err


def records(data):
    """[(name, sequence)] as ParseAll returns them (raises the parse error, if any)."""
    raw = bytes(data)
    seqs, offs, rec, err = pack(raw)
    out = []
    for i in range(len(offs) - 1):
        start = int(rec[i])
        end = raw.index(b"\n", start)
        out.append((raw[start + 1:end], seqs[int(offs[i]): int(offs[i + 1])].tobytes()))
    if err is not None:
        raise err
    return out


def workspace_bytes(nbytes: int) -> int:
    return int(_lib.lib().polyhip_fasta_workspace_bytes(nbytes))


def pack_dev(file_t, seqs_t, offsets_t, rec_start_t, result_t, work_t, max_records: int | None = None, stream=None):
    """Device-resident feeder on torch CUDA tensors; result_t int64[4] = (n, code, sequence bytes, headers)."""
    nbytes = file_t.numel()
    cap = nbytes // 2 + 2 if max_records is None else max_records
    _lib.check(_lib.lib().polyhip_fasta_pack_dev(
        file_t.data_ptr(), nbytes, seqs_t.data_ptr(), offsets_t.data_ptr(),
        rec_start_t.data_ptr() if rec_start_t is not None else None, cap, result_t.data_ptr(), work_t.data_ptr(),
        work_t.numel() * work_t.element_size(), _lib.stream_ptr(stream)))
