"""primers of bebop/poly on MI355X.

Mirrors primers/primers.go: ``SantaLucia`` (:70-105), ``MarmurDoty``
(:108-118), ``MeltingTemp`` (:121-128), plus the batch / scan entry points a
GPU needs.  All arithmetic runs in HIP (polyhip_santalucia_*,
polyhip_marmurdoty_*); this module only packs arguments.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .mash import _pack


def santalucia_batch_packed(seqs: np.ndarray, offsets: np.ndarray, primer_conc: float, salt_conc: float,
                            mg_conc: float):
    """Host-pointer entry point: (tm, dH, dS) float64[n] for a packed batch."""
    n = len(offsets) - 1
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    tm, dH, dS = (np.zeros(n, dtype=np.float64) for _ in range(3))
    _lib.check(_lib.lib().polyhip_santalucia_batch(seqs.ctypes.data, offsets.ctypes.data, n, primer_conc, salt_conc,
                                                   mg_conc, tm.ctypes.data, dH.ctypes.data, dS.ctypes.data))
    return tm, dH, dS


def SantaLucia(sequence, primerConcentration: float, saltConcentration: float, magnesiumConcentration: float):
    """primers.go:70-105 -> (meltingTemp, dH, dS)"""
    buf, offs = _pack([sequence])
    tm, dH, dS = santalucia_batch_packed(buf, offs, primerConcentration, saltConcentration, magnesiumConcentration)
    return float(tm[0]), float(dH[0]), float(dS[0])


def MeltingTemp(sequence) -> float:
    """primers.go:121-128"""
    return SantaLucia(sequence, 500e-9, 50e-3, 0.0)[0]


def marmurdoty_batch_packed(seqs: np.ndarray, offsets: np.ndarray) -> np.ndarray:
    n = len(offsets) - 1
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    tm = np.zeros(n, dtype=np.float64)
    _lib.check(_lib.lib().polyhip_marmurdoty_batch(seqs.ctypes.data, offsets.ctypes.data, n, tm.ctypes.data))
    return tm


def MarmurDoty(sequence) -> float:
    """primers.go:108-118"""
    buf, offs = _pack([sequence])
    return float(marmurdoty_batch_packed(buf, offs)[0])


def SantaLuciaBatch(seqs, primerConcentration=500e-9, saltConcentration=50e-3, magnesiumConcentration=0.0):
    """Additive batch API: SantaLucia for every sequence of a list."""
    buf, offs = _pack(seqs)
    return santalucia_batch_packed(buf, offs, primerConcentration, saltConcentration, magnesiumConcentration)


def SantaLuciaScan(genome, minLen: int, maxLen: int, primerConcentration=500e-9, saltConcentration=50e-3,
                   magnesiumConcentration=0.0):
    """Additive scan API: SantaLucia of genome[i:i+L] for every start i and every
    L in [minLen, maxLen].  Returns (tm, dH, dS), each float64[(maxLen-minLen+1), len-minLen+1];
    windows that run off the end are NaN."""
    buf, _ = _pack([genome])
    n = len(buf)
    nl = maxLen - minLen + 1
    ns = max(0, n - minLen + 1)
    tm, dH, dS = (np.zeros((max(nl, 0), ns), dtype=np.float64) for _ in range(3))
    _lib.check(_lib.lib().polyhip_santalucia_scan(buf.ctypes.data, n, minLen, maxLen, primerConcentration,
                                                  saltConcentration, magnesiumConcentration,
                                                  tm.ctypes.data, dH.ctypes.data, dS.ctypes.data))
    return tm, dH, dS


def SantaLuciaScanFirst(genome, minLen: int, maxLen: int, targetTm: float, primerConcentration=500e-9,
                        saltConcentration=50e-3, magnesiumConcentration=0.0):
    """Additive scan API: for every start of `genome` the first length in [minLen, maxLen] whose SantaLucia Tm is not
    below targetTm (the grow loop of pcr.go:47-53 at every position) -> (first_len uint16[len - minLen + 1] with 0 =
    none, first_tm float64, NaN where none).  Only 10 bytes per start cross PCIe."""
    buf, _ = _pack([genome])
    n = len(buf)
    ns = max(0, n - minLen + 1)
    first_len = np.zeros(ns, dtype=np.uint16)
    first_tm = np.full(ns, np.nan, dtype=np.float64)
    _lib.check(_lib.lib().polyhip_santalucia_scan_first(buf.ctypes.data, n, minLen, maxLen, primerConcentration,
                                                        saltConcentration, magnesiumConcentration, targetTm,
                                                        first_len.ctypes.data, first_tm.ctypes.data))
    return first_len, first_tm


def santalucia_scan_first_dev(seq_t, length: int, start0: int, nstarts: int, minLen: int, maxLen: int, primer_conc: float,
                              salt_conc: float, mg_conc: float, target_tm: float, first_len_t, first_tm_t=None, stream=None) -> None:
    """Device-resident reduced scan on torch CUDA tensors (uint8 genome, int16/uint16 first_len, float64 first_tm)."""
    assert seq_t.is_cuda and first_len_t.is_cuda and first_len_t.element_size() == 2 and first_len_t.numel() >= nstarts
    _lib.check(_lib.lib().polyhip_santalucia_scan_first_dev(
        seq_t.data_ptr(), length, start0, nstarts, minLen, maxLen, primer_conc, salt_conc, mg_conc, target_tm,
        first_len_t.data_ptr(), first_tm_t.data_ptr() if first_tm_t is not None else None, _lib.stream_ptr(stream)))


def santalucia_scan_dev(seq_t, length: int, start0: int, nstarts: int, minLen: int, maxLen: int,
                        primer_conc: float, salt_conc: float, mg_conc: float, tm_t, dH_t, dS_t, ld: int,
                        stream=None) -> None:
    """Device-resident scan on torch CUDA tensors (uint8 genome, float64 planes)."""
    assert seq_t.is_cuda and tm_t.is_cuda and dH_t.is_cuda and dS_t.is_cuda
    need = (maxLen - minLen) * ld + nstarts  # the last plane only has to hold its nstarts values
    assert tm_t.numel() >= need and dH_t.numel() >= need and dS_t.numel() >= need
    assert tm_t.element_size() == 8 and seq_t.element_size() == 1 and seq_t.numel() >= length
    _lib.check(_lib.lib().polyhip_santalucia_scan_dev(
        seq_t.data_ptr(), length, start0, nstarts, minLen, maxLen, primer_conc, salt_conc, mg_conc,
        tm_t.data_ptr(), dH_t.data_ptr(), dS_t.data_ptr(), ld, _lib.stream_ptr(stream)))
