"""The N > 1 path on CPU: world_size-2 (and 3) gloo process groups run the SAME sharding /
all-gather plumbing bench.py uses on RCCL (poly_amd/sharding.py), with the CPU oracle standing
in for the HIP kernels as the compute callable.  Checks: shards tile the units exactly, the
gathered sketch set is identical on every rank (equal and ragged shard sizes), and the
row blocks stack to the single-process all-vs-all matrix; K4's start slices with halo
reproduce the whole-genome scan."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as orc
from poly_amd import sharding


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _sketches(n, s, seed):
    buf = orc.synth_dna(seed, n * 600)
    # make neighbours related so the matrix is not trivially diagonal
    b = buf.reshape(n, 600).copy()
    b[1::2, :400] = b[0::2, :400][: len(b[1::2])]
    offs = np.arange(0, (n + 1) * 600, 600, dtype=np.uint64)
    return orc.mash_sketch_batch(b.reshape(-1), offs, 15, s)


def _oracle_counts(X, Y):
    X = X.numpy().view(np.uint32)
    Y = Y.numpy().view(np.uint32)
    out = np.zeros((len(X), len(Y)), np.int16)
    for i in range(len(X)):
        for j in range(len(Y)):
            out[i, j] = orc.mash_shared(np.ascontiguousarray(X[i]), np.ascontiguousarray(Y[j]))
    return torch.from_numpy(out)


def _worker(rank, world, port, n_total, ragged, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S = torch.from_numpy(_sketches(n_total, 40, 0xC3).view(np.int32))
        if ragged:
            cuts = [0, 3, n_total] if world == 2 else [0, 2, 9, n_total]
            lo, hi = cuts[rank], cuts[rank + 1]
        else:
            lo, hi = sharding.shard_range(n_total, rank, world)
        local = S[lo:hi].clone()
        counts, row0, gathered = sharding.allvsall_row_block(local, _oracle_counts)
        assert row0 == lo and torch.equal(gathered, S)
        q.put((rank, row0, counts.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,ragged", [(2, False), (2, True), (3, True)])
def test_allgather_row_blocks_stack_to_full_matrix(world, ragged):
    n_total = 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, ragged, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort()
    S = torch.from_numpy(_sketches(n_total, 40, 0xC3).view(np.int32))
    whole = _oracle_counts(S, S).numpy()
    assert (np.vstack([g[2] for g in got]) == whole).all()
    assert (whole.diagonal() == 40).all() and (whole[0, 1] > 0)


def test_shard_range_tiles_exactly():
    for n in (0, 1, 7, 8, 100, 1_000_003):
        for world in (1, 2, 3, 8):
            edges = [sharding.shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def test_scan_shards_with_halo_reproduce_whole_scan():
    """K4's multi-GPU partition: rank r scans starts [start0, start0+nstarts) of the SAME genome buffer
    (its Lmax-1 byte halo is just the following bytes); stitched planes equal the single-rank scan."""
    g = bytes(orc.synth_dna(0xC5, 400))
    Lmin, Lmax = 18, 30

    def scan(start0, nstarts):
        out = np.full((Lmax - Lmin + 1, nstarts), np.nan)
        for L in range(Lmin, Lmax + 1):
            for i in range(start0, start0 + nstarts):
                if i + L <= len(g):
                    out[L - Lmin, i - start0] = orc.santalucia(g[i:i + L], 500e-9, 50e-3, 0.0)[0]
        return out

    whole = scan(0, len(g) - Lmin + 1)
    for world in (2, 3):
        parts = [scan(*sharding.scan_shard(len(g), Lmin, r, world)) for r in range(world)]
        st = np.hstack(parts)
        assert st.shape == whole.shape
        assert ((st == whole) | (np.isnan(st) & np.isnan(whole))).all()
