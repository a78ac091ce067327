"""Synthetic workloads of SURVEY.md 8d, host side (numpy), bit-reproducible.

The reference's own generator (random.DNASequence, random/random.go:52-63) draws from Go's
math/rand and cannot be reproduced outside Go, so SURVEY 8d defines the inputs on
splitmix64: base i of a stream is "ACGT"[(x >> 2*(i%32)) & 3] with x the (i/32)-th output.
splitmix64 is counter based (output j = mix(seed + (j+1)*gamma)), so any slice of a stream
can be produced without the ones before it, and numpy can produce it vectorised.

bench.py / bench_extra.py, the tests and the oracle-side CPU baselines all take their inputs
from here so that every number is quoted on the same bytes.
"""
from __future__ import annotations

import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def splitmix64(seed: int, first: int, count: int) -> np.ndarray:
    """outputs first .. first+count-1 of the splitmix64 stream seeded with `seed` (uint64[count])"""
    with np.errstate(over="ignore"):
        j = np.arange(first + 1, first + 1 + count, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + j * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def synth_dna(seed: int, n: int, first: int = 0) -> np.ndarray:
    """bases first .. first+n-1 of the synthetic DNA stream (same bytes as polyhip_synth_dna_dev)"""
    w0, w1 = first // 32, (first + n + 31) // 32
    x = splitmix64(seed, w0, w1 - w0)
    sh = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
    codes = ((x[:, None] >> sh) & np.uint64(3)).astype(np.uint8).reshape(-1)
    return _ACGT[codes[first - 32 * w0: first - 32 * w0 + n]]


# ---- configs[3]: Smith-Waterman reads ---------------------------------------------

C4_REF_SEED = 0xC4
C4_READ_SEED = 0xC4 + 1
C4_SLACK = 32  # source bases drawn beyond the read length, to feed deletions


def config4_reference(LB: int = 5000) -> np.ndarray:
    return synth_dna(C4_REF_SEED, LB)


def _splitmix64_torch(seed: int, first: int, count: int, device):
    """splitmix64 outputs first .. first+count-1 as int64 bit patterns (torch has no uint64 arithmetic:
    two's-complement multiply wraps identically, logical shifts are arithmetic shifts + mask)"""
    import torch

    def i64(v: int) -> int:
        v &= 0xFFFFFFFFFFFFFFFF
        return v - (1 << 64) if v >> 63 else v

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)

    j = torch.arange(first + 1, first + 1 + count, dtype=torch.int64, device=device)
    z = j * i64(0x9E3779B97F4A7C15) + i64(seed)
    z = (z ^ lsr(z, 30)) * i64(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * i64(0x94D049BB133111EB)
    return z ^ lsr(z, 31)


def config4_reads(n: int, LA: int = 150, LB: int = 5000, first: int = 0, sub: float = 0.05, indel: float = 0.01,
                  device=None):
    """(reference uint8[LB], reads uint8[n][LA]): read r (global index first + r) = a random window of the
    reference pushed through a per-base mutation channel -- substitution by one of the three OTHER bases
    with probability `sub`, deletion with `indel`/2, insertion of a random base in front with `indel`/2 --
    and cut to exactly LA bases.  One splitmix64 output per (read, source base): bits 0..23 pick the event,
    24..25 the substituted base (offset 1..3), 26..27 the inserted base; output (LA+SLACK) of a read's block
    picks the window start.  Reads depend only on their global index, so shards of the 1M-read batch are
    slices of one definition.  Integer torch ops only (identical on CPU and GPU): `device=None` returns numpy
    arrays computed on the CPU, a CUDA device returns CUDA tensors."""
    import torch
    dev = torch.device("cpu") if device is None else device
    ref_np = config4_reference(LB)
    W = LA + C4_SLACK
    per = W + 1
    refc = torch.from_numpy(np.searchsorted(_ACGT, ref_np).astype(np.int64)).to(dev)  # codes 0..3
    acgt = torch.from_numpy(_ACGT.copy()).to(dev)
    t_sub = int(sub * (1 << 24))
    t_del = t_sub + int(indel / 2 * (1 << 24))
    t_ins = t_del + int(indel / 2 * (1 << 24))
    reads = torch.empty((n, LA), dtype=torch.uint8, device=dev)
    cols = torch.arange(W, device=dev)[None, :]
    step = 100_000
    for c0 in range(0, n, step):
        m = min(step, n - c0)
        x = _splitmix64_torch(C4_READ_SEED, (first + c0) * per, m * per, dev).view(m, per)
        start = ((x[:, W] >> 1) & 0x7FFFFFFFFFFFFFFF) % (LB - W)  # top 63 bits: non-negative
        src = refc[start[:, None] + cols]  # (m, W) codes
        xs = x[:, :W]
        ev = xs & 0xFFFFFF
        o_sub = ((xs >> 24) & 3) % 3 + 1
        b_ins = (xs >> 26) & 3
        is_sub = ev < t_sub
        is_del = (ev >= t_sub) & (ev < t_del)
        is_ins = (ev >= t_del) & (ev < t_ins)
        base = torch.where(is_sub, (src + o_sub) & 3, src)
        emit = 1 - is_del.long() + is_ins.long()  # output bases this source base yields: 0 / 1 / 2
        pos = torch.cumsum(emit, dim=1) - emit  # first output slot of each source base
        if int((pos[:, -1] + emit[:, -1]).min()) < LA:
            raise RuntimeError("config4_reads: a read ran out of source bases (raise C4_SLACK)")
        dump = 2 * W + 1
        buf = torch.zeros((m, 2 * W + 2), dtype=torch.int64, device=dev)
        # inserted base first, then the source base itself; slots are distinct, so scatter order is irrelevant
        buf.scatter_(1, torch.where(is_ins, pos, dump), b_ins)
        buf.scatter_(1, torch.where(is_del, dump, pos + is_ins.long()), base)
        reads[c0:c0 + m] = acgt[buf[:, :LA]]
    if device is None:
        return ref_np, reads.numpy()
    return torch.from_numpy(ref_np.copy()).to(dev), reads
