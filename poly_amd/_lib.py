"""ctypes loader for libpolyhip.so -- the only bridge between the Python host
layer and the HIP kernels.  There is no fallback: if the library is missing or
a call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# POLYHIP_LIB: another build of the same library (kernel-variant A/B runs, scripts/build_variant.sh); tests use the default
LIB_PATH = os.environ.get("POLYHIP_LIB") or os.path.join(HERE, "libpolyhip.so")

OK, ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_PANIC, ERR_SYMBOL = 0, -1, -2, -3, -4, -5


class PolyhipError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"polyhip status {status}: {message}")
        self.status = status
        self.message = message


class GoPanic(PolyhipError):
    """The reference (Go) would panic on these arguments."""


_lib = None

# name -> (restype, argtypes); mirrors include/polyhip.h one to one
_u64, _u32, _vp, _i32, _dbl = C.c_uint64, C.c_uint32, C.c_void_p, C.c_int32, C.c_double
SIGNATURES = {
    "polyhip_abi_version": (C.c_int, []),
    "polyhip_last_error": (C.c_char_p, []),
    "polyhip_device_count": (C.c_int, []),
    "polyhip_set_device": (C.c_int, [C.c_int]),
    "polyhip_device_arch": (C.c_int, [C.c_char_p, C.c_size_t]),
    "polyhip_set_devices": (C.c_int, [_vp, C.c_int]),
    "polyhip_get_devices": (C.c_int, [_vp, C.c_int]),
    "polyhip_init": (C.c_int, [C.c_int]),
    "polyhip_shutdown": (C.c_int, []),
    "polyhip_synth_dna_dev": (C.c_int, [_u64, _u64, _vp, _u64, _vp]),
    "polyhip_mash_sketch_batch": (C.c_int, [_vp, _vp, _u64, _u32, _u32, _vp]),
    "polyhip_mash_sketch_batch_dev": (C.c_int, [_vp, _vp, _u64, _u32, _u32, _vp, _vp]),
    "polyhip_mash_shared_counts_workspace_bytes": (C.c_size_t, [_u64, _u32, _u64, _u32]),
    "polyhip_mash_shared_counts_dev": (C.c_int, [_vp, _u64, _u32, _vp, _u64, _u32, _vp, _u64, _vp, C.c_size_t, _vp]),
    "polyhip_mash_index_build_dev": (C.c_int, [_vp, _u64, _u32, _vp, C.c_size_t, _vp]),
    "polyhip_mash_index_build_part_dev": (C.c_int, [_vp, _u64, _u32, _u32, _u32, _vp, C.c_size_t, _vp]),
    "polyhip_mash_index_part_spans": (C.c_int, [_u64, _u32, _u32, _vp, C.c_size_t, _vp, _vp, _vp]),
    "polyhip_mash_index_finalize_dev": (C.c_int, [_u64, _u32, _vp, C.c_size_t, _vp]),
    "polyhip_mash_index_format_dev": (C.c_int, [_vp, _vp]),
    "polyhip_mash_index_build_info_dev": (C.c_int, [_vp, _vp]),
    "polyhip_mash_index_allgather_dev": (C.c_int, [_vp, _u64, _u32, _vp, C.c_size_t, _vp]),
    "polyhip_mash_shared_counts_reuse_dev": (C.c_int, [_vp, _u64, _u32, _vp, _u64, _u32, _vp, _u64, _vp, C.c_size_t, _vp]),
    "polyhip_mash_shared_counts_mode_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "polyhip_mash_distance_from_counts_dev": (C.c_int, [_vp, _u64, _u64, _u64, _u32, _u32, _vp, _u64, _vp]),
    "polyhip_mash_distance_matrix": (C.c_int, [_vp, _u64, _u32, _vp, _u64, _u32, _vp, _vp]),
    "polyhip_mash_sketch_distance_matrix": (C.c_int, [_vp, _vp, _u64, _u32, _u32, _vp, _vp, _vp]),
    "polyhip_mash_sketch_distance_matrix_last_path": (C.c_int, []),
    "polyhip_mash_sketch_distance_matrix_last_info": (C.c_int, [_vp]),
    "polyhip_scoring_create": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.POINTER(C.c_void_p)]),
    "polyhip_scoring_destroy": (C.c_int, [_vp]),
    "polyhip_sw_workspace_bytes": (C.c_size_t, [_vp, _u64, _u32, _u64, C.c_int]),
    "polyhip_sw_batch_dev": (C.c_int, [_vp, _vp, _vp, _u64, _u32, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp,
                                       C.c_size_t, _vp]),
    "polyhip_sw_batch": (C.c_int, [_vp, _vp, _vp, _u64, _vp, _vp, _u64, _vp, _vp, _vp, _vp]),
    "polyhip_sw_last_path": (C.c_int, []),
    "polyhip_sw_last_packed_half": (C.c_int, []),
    "polyhip_sw_last_packed_lanes": (C.c_int, []),
    "polyhip_sw_traceback_last_path": (C.c_int, []),
    "polyhip_sw_traceback_last_half": (C.c_int, []),
    "polyhip_nw_last_path": (C.c_int, []),
    "polyhip_nw_workspace_bytes": (C.c_size_t, [_u64, _u32, _u64]),
    "polyhip_nw_align_batch_dev": (C.c_int, [_vp, _vp, _vp, _u64, _u32, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp, _u32, _vp,
                                             C.c_size_t, _vp]),
    "polyhip_nw_align_batch": (C.c_int, [_vp, _vp, _vp, _u64, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp, _u32]),
    "polyhip_sw_traceback_stride": (C.c_uint32, [_vp, _u32, _u64]),
    "polyhip_sw_traceback_workspace_bytes": (C.c_size_t, [_vp, _u64, _u32, _u64]),
    "polyhip_sw_traceback_dev": (C.c_int, [_vp, _vp, _vp, _u64, _u32, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                           _u32, _vp, C.c_size_t, _vp]),
    "polyhip_sw_align_batch_dev": (C.c_int, [_vp, _vp, _vp, _u64, _u32, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32,
                                             _vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    "polyhip_sw_align_batch": (C.c_int, [_vp, _vp, _vp, _u64, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32]),
    "polyhip_sw_align_batch_packed": (C.c_int, [_vp, _vp, _vp, _u64, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u64]),
    "polyhip_santalucia_scan_dev": (C.c_int, [_vp, _u64, _u64, _u64, _u32, _u32, _dbl, _dbl, _dbl, _vp, _vp, _vp,
                                              _u64, _vp]),
    "polyhip_santalucia_scan": (C.c_int, [_vp, _u64, _u32, _u32, _dbl, _dbl, _dbl, _vp, _vp, _vp]),
    "polyhip_santalucia_scan_first_dev": (C.c_int, [_vp, _u64, _u64, _u64, _u32, _u32, _dbl, _dbl, _dbl, _dbl, _vp, _vp, _vp]),
    "polyhip_santalucia_scan_first": (C.c_int, [_vp, _u64, _u32, _u32, _dbl, _dbl, _dbl, _dbl, _vp, _vp]),
    "polyhip_santalucia_batch_dev": (C.c_int, [_vp, _vp, _u64, _dbl, _dbl, _dbl, _vp, _vp, _vp, _vp]),
    "polyhip_santalucia_batch": (C.c_int, [_vp, _vp, _u64, _dbl, _dbl, _dbl, _vp, _vp, _vp]),
    "polyhip_marmurdoty_batch_dev": (C.c_int, [_vp, _vp, _u64, _vp, _vp]),
    "polyhip_marmurdoty_batch": (C.c_int, [_vp, _vp, _u64, _vp]),
    "polyhip_least_rotation_batch_dev": (C.c_int, [_vp, _vp, _u64, _u64, _vp, _vp, _vp]),
    "polyhip_least_rotation_batch": (C.c_int, [_vp, _vp, _u64, _vp, _vp]),
    "polyhip_seqhash_workspace_bytes": (C.c_size_t, [_u64, _u64, C.c_int, C.c_int]),
    "polyhip_seqhash_batch_dev": (C.c_int, [_vp, _vp, _u64, _u64, _u64, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp,
                                            C.c_size_t, _vp]),
    "polyhip_seqhash_batch": (C.c_int, [_vp, _vp, _u64, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "polyhip_fastq_workspace_bytes": (C.c_size_t, [_u64]),
    "polyhip_fastq_pack_dev": (C.c_int, [_vp, _u64, _vp, _vp, _vp, _u64, _vp, _vp, C.c_size_t, _vp]),
    "polyhip_fastq_pack": (C.c_int, [_vp, _u64, _vp, _vp, _vp, _u64, _vp]),
    "polyhip_fasta_workspace_bytes": (C.c_size_t, [_u64]),
    "polyhip_fasta_pack_dev": (C.c_int, [_vp, _u64, _vp, _vp, _vp, _u64, _vp, _vp, C.c_size_t, _vp]),
    "polyhip_fasta_pack": (C.c_int, [_vp, _u64, _vp, _vp, _vp, _u64, _vp]),
    "polyhip_comm_unique_id": (C.c_int, [_vp]),
    "polyhip_comm_init_rank": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "polyhip_comm_destroy": (C.c_int, [_vp]),
    "polyhip_comm_rank": (C.c_int, [_vp]),
    "polyhip_comm_size": (C.c_int, [_vp]),
    "polyhip_allgather_sketches_dev": (C.c_int, [_vp, _vp, _u64, _u32, _vp, _vp]),
    "polyhip_allgatherv_dev": (C.c_int, [_vp, _vp, _vp, _vp]),
}


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m poly_amd.build` "
                "(poly_amd has no CPU fallback)")
        # PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64 (same
        # SONAME as /opt/rocm's).  Two HIP runtimes in one process cannot both own
        # the GPU, so torch's copy must be mapped first; libpolyhip.so then binds
        # to it by SONAME.  (A torch-free consumer -- the cgo shim -- just gets
        # /opt/rocm/lib/libamdhip64.so.7.)
        try:
            import torch  # noqa: F401
        except ImportError:  # pragma: no cover - torch is present in this image
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError if the symbol is not exported
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def check(status: int) -> None:
    if status == OK:
        return
    msg = lib().polyhip_last_error().decode("utf-8", "replace")
    if status == ERR_PANIC:
        raise GoPanic(status, msg)
    raise PolyhipError(status, msg)


def stream_ptr(stream=None) -> int:
    """hipStream_t of a torch stream (default: torch's current stream)."""
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return int(s.cuda_stream)
