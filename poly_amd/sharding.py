"""Multi-GPU layout of the hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

SURVEY 8e: every kernel shards by independent units with no data-path
collective (reads for K1, pairs for K3, start positions for K4, sequences for
K5); the single exchange step of the path is the all-gather of per-rank
sketches in front of the all-vs-all distance matrix (K2), after which each
rank computes its own row block.  This module holds only that plumbing; the
compute callables are the HIP entry points (the tests pass a checker instead,
so the partition/gather logic runs under gloo on CPU).
"""
from __future__ import annotations


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of n independent units for `rank`: the first
    n % world ranks get one extra unit."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def scan_shard(length: int, min_len: int, rank: int, world: int) -> tuple[int, int]:
    """K4: (start0, nstarts) of this rank's slice of the len - min_len + 1 window
    starts; the (max_len - 1)-byte right halo is read from the same genome buffer."""
    nstarts = max(0, length - min_len + 1)
    lo, hi = shard_range(nstarts, rank, world)
    return lo, hi - lo


def gather_sketches(local, group=None):
    """R1: all-gather of the per-rank sketch blocks (n_local, s) -> (N, s) on every
    rank, plus this rank's first row.  Ranks may hold different n_local (blocks are
    padded to the maximum for the collective and trimmed afterwards)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local, 0
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(t.item()) for t in sizes]
    nmax = max(sizes)
    if all(sz == nmax for sz in sizes):
        out = torch.empty((world * nmax, local.shape[1]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)  # one ncclAllGather
        return out, rank * nmax
    padded = torch.zeros((nmax, local.shape[1]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    buf = torch.empty((world * nmax, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    out = torch.cat([buf[r * nmax: r * nmax + sizes[r]] for r in range(world)])
    return out, sum(sizes[:rank])


def allvsall_row_block(local_sketches, compute_counts, group=None):
    """All-vs-all shared-hash counts, sharded by rows: gathers every rank's sketches
    and returns (counts for this rank's rows x all N columns, row0, gathered).
    `compute_counts(X, Y) -> counts` is poly_amd.mash's K2 on the GPU box."""
    gathered, row0 = gather_sketches(local_sketches, group)
    X = gathered[row0: row0 + local_sketches.shape[0]]
    return compute_counts(X, gathered), row0, gathered
