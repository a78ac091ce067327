"""The library's device list: one host-pointer call spread over several GPUs of the node.

SURVEY 8b (``polyhip_init(n_devices)``) / 8e: with a list of n devices every host-pointer entry point (what the Go
wrappers call: ``mash.SketchBatch``, ``mash.DistanceMatrix``, ``align.SmithWatermanBatch``, ``primers.SantaLuciaScan``,
``seqhash.HashBatch`` ...) cuts its batch into n contiguous shards and runs shard q on device ``ids[q]``.  An id may
repeat (``[0, 0, 0]``): the shards then share that GPU, which is how the fan-out is tested on a one-GPU box.
"""
from __future__ import annotations

import contextlib
import ctypes as C

from . import _lib


def set_devices(ids) -> None:
    """polyhip_set_devices: ``[]`` / ``None`` = back to the calling thread's current device."""
    ids = list(ids or [])
    arr = (C.c_int * max(1, len(ids)))(*ids)
    _lib.check(_lib.lib().polyhip_set_devices(C.addressof(arr), len(ids)))


def get_devices() -> list[int]:
    arr = (C.c_int * 64)()
    n = _lib.lib().polyhip_get_devices(C.addressof(arr), 64)
    return [arr[i] for i in range(min(n, 64))]


def init(n_devices: int = 0) -> None:
    """polyhip_init: devices 0 .. n-1 (n <= 0: every visible device)."""
    _lib.check(_lib.lib().polyhip_init(int(n_devices)))


def shutdown() -> None:
    _lib.check(_lib.lib().polyhip_shutdown())


@contextlib.contextmanager
def devices(ids):
    """``with devices([0, 0, 0]): ...`` -- the list for the duration of a block, the previous one afterwards."""
    before = get_devices()
    set_devices(ids)
    try:
        yield
    finally:
        set_devices(before)
