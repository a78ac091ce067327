"""R1 through the C ABI: the communicator a torch-free host (the Go/cgo drop-in) uses -- polyhip_comm_* and the
collectives of csrc/comm.hip (RCCL over xGMI, resolved with dlopen).  The rendezvous (handing rank 0's 128-byte id to
the other ranks) is the host's business: the callers here pass it through whatever channel they have (bench.py: one
torch.distributed broadcast; tests: a file)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def unique_id() -> bytes:
    """rank 0 only: the 128-byte id every rank's `Comm` needs"""
    buf = (C.c_uint8 * 128)()
    _lib.check(_lib.lib().polyhip_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """one rank's communicator on the calling thread's current HIP device (polyhip_comm_init_rank)"""

    def __init__(self, uid: bytes, rank: int, nranks: int):
        assert len(uid) == 128
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        _lib.check(_lib.lib().polyhip_comm_init_rank(buf, rank, nranks, C.byref(h)))
        self._h = h
        self.rank, self.nranks = rank, nranks

    def close(self) -> None:
        if self._h is not None:
            h, self._h = self._h, None
            _lib.check(_lib.lib().polyhip_comm_destroy(h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def allgather_sketches(self, local_t, all_t, stream=None) -> None:
        """polyhip_allgather_sketches_dev: all_t (nranks * n_local, s) <- every rank's local_t (n_local, s); the SAME
        n_local on every rank (pad ragged shards, see sharding.gather_sketches)."""
        n_local, s = local_t.shape
        assert local_t.is_cuda and all_t.is_cuda and local_t.is_contiguous() and all_t.is_contiguous()
        assert local_t.element_size() == 4 and all_t.element_size() == 4 and all_t.numel() == self.nranks * n_local * s
        _lib.check(_lib.lib().polyhip_allgather_sketches_dev(self._h, local_t.data_ptr(), n_local, s, all_t.data_ptr(),
                                                             _lib.stream_ptr(stream)))

    def allgatherv(self, buf_t, offsets, stream=None) -> None:
        """polyhip_allgatherv_dev: in place, rank r owns bytes [offsets[r], offsets[r+1]) of buf_t"""
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        assert len(off) == self.nranks + 1 and buf_t.is_cuda and buf_t.is_contiguous()
        _lib.check(_lib.lib().polyhip_allgatherv_dev(self._h, buf_t.data_ptr(), off.ctypes.data, _lib.stream_ptr(stream)))

    def index_allgather(self, ny: int, sy: int, work_t, stream=None) -> None:
        """polyhip_mash_index_allgather_dev: exchange the parts of the index every rank built with
        mash.index_build_part_dev(part = rank, nparts = nranks) and finish it"""
        _lib.check(_lib.lib().polyhip_mash_index_allgather_dev(self._h, ny, sy, work_t.data_ptr(),
                                                               work_t.numel() * work_t.element_size(),
                                                               _lib.stream_ptr(stream)))
