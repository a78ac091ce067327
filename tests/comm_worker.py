"""One rank of the multi-process test of R1 THROUGH THE C ABI (tests/test_comm_gpu.py starts N of these, one per GPU):
polyhip_comm_unique_id / _init_rank (the 128-byte id travels through a file, as a Go driver would use a pipe),
polyhip_allgather_sketches_dev, polyhip_mash_index_build_part_dev + polyhip_mash_index_allgather_dev, and the row
block of the all-vs-all against the index assembled from the ranks' parts.  No torch.distributed anywhere: torch is
only the device allocator here.

    python tests/comm_worker.py RANK NRANKS IDFILE OUTDIR [DEVICE]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rank_sketches(rank: int, nfam: int, copies: int, s: int) -> np.ndarray:
    """families of related ascending sketches, a pure function of the rank (every process can rebuild every shard)"""
    rng = np.random.default_rng(1000 + rank)
    out = []
    for _ in range(nfam):
        base = rng.integers(0, 1 << 27, s, dtype=np.uint32)
        for _ in range(copies):
            m = base.copy()
            hit = rng.random(s) < 0.15
            m[hit] = rng.integers(0, 1 << 27, int(hit.sum()), dtype=np.uint32)
            m.sort()
            out.append(m)
    return np.stack(out)


def main() -> int:
    rank, nranks, idfile, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    device = int(sys.argv[5]) if len(sys.argv) > 5 else rank
    import torch
    from poly_amd import comm, mash

    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    if rank == 0:
        uid = comm.unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 120:
                raise TimeoutError("rank 0 never wrote the communicator id")
            time.sleep(0.05)
        uid = open(idfile, "rb").read()
    c = comm.Comm(uid, rank, nranks)
    nfam, copies, s = 12, 10, 256
    local = torch.from_numpy(rank_sketches(rank, nfam, copies, s).view(np.int32)).to(dev)
    n_local = local.shape[0]
    gathered = torch.zeros((nranks * n_local, s), dtype=torch.int32, device=dev)
    c.allgather_sketches(local, gathered)
    torch.cuda.synchronize()
    want = np.concatenate([rank_sketches(r, nfam, copies, s) for r in range(nranks)])
    ok_gather = bool((gathered.cpu().numpy().view(np.uint32) == want).all())
    # the index in parts: mine, then everybody's through the ragged all-gather
    N = gathered.shape[0]
    wb = mash.shared_counts_workspace_bytes(n_local, s, N, s)
    work = torch.zeros(wb, dtype=torch.uint8, device=dev)
    mash.index_build_part_dev(gathered, rank, nranks, work)
    c.index_allgather(N, s, work)
    X = gathered[rank * n_local:(rank + 1) * n_local]
    counts = torch.zeros((n_local, N), dtype=torch.int16, device=dev)
    mash.shared_counts_reuse_dev(X, gathered, counts, work)
    # the same row block from an index this rank built alone
    work1 = torch.zeros(wb, dtype=torch.uint8, device=dev)
    counts1 = torch.zeros_like(counts)
    mash.shared_counts_dev(X, gathered, counts1, work1)
    torch.cuda.synchronize()
    ok_counts = bool(torch.equal(counts, counts1))
    np.save(os.path.join(outdir, f"counts_{rank}.npy"), counts.cpu().numpy().view(np.uint16))
    c.close()
    with open(os.path.join(outdir, f"rank_{rank}.json"), "w") as f:
        json.dump({"rank": rank, "nranks": nranks, "device": torch.cuda.get_device_name(device), "gather_ok": ok_gather,
                   "counts_equal_local_index": ok_counts, "n_local": n_local, "N": N}, f)
    return 0 if ok_gather and ok_counts else 1


if __name__ == "__main__":
    sys.exit(main())
