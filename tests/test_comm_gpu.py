"""R1 and the multi-rank index through the C ABI (csrc/comm.hip, polyhip_mash_index_build_part_dev /
_allgather_dev): what a torch-free Go host calls for BASELINE configs[2].

One GPU is enough for: the index built in parts equals the one-shot index (same bucket starts, same items per bucket)
and its counts equal the oracle's merge loop (mash.go:107-135); the collectives on a 1-rank communicator.  With two or
more GPUs visible the same calls run in N PROCESSES (tests/comm_worker.py, tests/abi/abi_allgather: no
torch.distributed, the id travels through a file / a pipe); with one GPU those tests skip."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _families(rng, nfam, copies, s, hi=1 << 28, sub=0.12):
    out = []
    for _ in range(nfam):
        base = rng.integers(0, hi, s, dtype=np.uint32)
        for _ in range(copies):
            m = base.copy()
            hit = rng.random(s) < sub
            m[hit] = rng.integers(0, hi, int(hit.sum()), dtype=np.uint32)
            m.sort()
            out.append(m)
    return np.stack(out)


def _index_view(mash, work, ny, sy):
    """(start[0..nbk], items (n, 2) uint32) of the index in a workspace, read through polyhip_mash_index_part_spans"""
    it, st = mash.index_part_spans(ny, sy, 1, work)
    w = work.cpu().numpy()
    start = w[int(st[0]): int(st[1]) + 4].view(np.uint32)          # nbk + 1 entries
    raw = w[int(it[0]): int(it[1])].view(np.uint32)
    if mash.index_item_bytes(work) == 8:                           # (value, id | occurrence) -> one 64-bit key per item
        raw = raw.reshape(-1, 2)
        return start, raw[:, 0].astype(np.uint64) << 32 | raw[:, 1]
    return start, raw.astype(np.uint64)                            # compact 4-byte items


@pytest.mark.parametrize("nparts", [2, 3, 8])
@pytest.mark.parametrize("shape", [(40, 25, 200), (3, 7, 1000), (300, 4, 64)])
def test_index_parts_equal_one_shot_and_oracle(nparts, shape):
    import torch
    from poly_amd import mash
    nfam, copies, s = shape
    rng = np.random.default_rng(nparts * 1000 + s)
    S = _families(rng, nfam, copies, s)
    S[1, : s // 2] = S[1, 0]                 # a run of equal hashes (occurrence numbers)
    S[-1] = rng.integers(0, 1 << 28, s)      # one unsorted (irregular) sketch: the merge loop's pairs
    dev = torch.device("cuda:0")
    Y = torch.from_numpy(S.view(np.int32)).to(dev)
    N = Y.shape[0]
    wb = mash.shared_counts_workspace_bytes(N, s, N, s)
    one = torch.zeros(wb, dtype=torch.uint8, device=dev)
    mash.index_build_dev(Y, one)
    parts = torch.zeros(wb, dtype=torch.uint8, device=dev)
    for p in range(nparts):                   # what ranks 0 .. nparts-1 would each do, one after the other
        mash.index_build_part_dev(Y, p, nparts, parts)
    mash.index_finalize_dev(N, s, parts)
    torch.cuda.synchronize()
    it, st = mash.index_part_spans(N, s, nparts, parts)
    assert it[0] < it[-1] and all(it[p] <= it[p + 1] for p in range(nparts)) and all(st[p] <= st[p + 1] for p in range(nparts))
    sizes = np.diff(it.astype(np.int64)) // mash.index_item_bytes(parts)
    assert sizes.sum() == (N - 1) * s          # every item of the regular sketches is in exactly one part
    if N * s >= 20000:
        assert sizes.max() <= 2.0 * sizes.sum() / nparts + 4096   # parts are balanced by items, not by value range
    s1, i1 = _index_view(mash, one, N, s)
    s2, i2 = _index_view(mash, parts, N, s)
    assert (s1 == s2).all(), "bucket starts of the assembled index differ from the one-shot index"
    # same items in every bucket (the order inside a bucket is whatever the atomics gave)
    assert len(i1) == (N - 1) * s and mash.index_item_bytes(one) == mash.index_item_bytes(parts)
    for b in range(0, len(s1) - 1, max(1, (len(s1) - 1) // 997)):      # sampled buckets: the same items, any order
        assert (np.sort(i1[s1[b]:s1[b + 1]]) == np.sort(i2[s2[b]:s2[b + 1]])).all()
    assert (np.sort(i1) == np.sort(i2)).all()
    assert mash.shared_counts_mode(one)[4] == mash.shared_counts_mode(parts)[4]   # self-join size recomputed
    counts = torch.zeros((N, N), dtype=torch.int16, device=dev)
    mash.shared_counts_reuse_dev(Y, Y, counts, parts)
    torch.cuda.synchronize()
    got = counts.cpu().numpy().view(np.uint16)
    rows = sorted(set(rng.integers(0, N, 12).tolist() + [0, 1, N - 1]))
    for i in rows:
        for j in range(N):
            assert got[i, j] == orc.mash_shared(S[i], S[j]), (i, j)


def test_comm_one_rank_collectives():
    """unique id -> init -> all-gather -> ragged all-gather -> index all-gather on a 1-rank communicator: every RCCL
    entry point the N-rank path uses is resolved and called on this box"""
    import torch
    from poly_amd import comm, mash
    dev = torch.device("cuda:0")
    c = comm.Comm(comm.unique_id(), 0, 1)
    local = torch.randint(0, 1 << 31, (300, 64), dtype=torch.int32, device=dev)
    out = torch.zeros_like(local)
    c.allgather_sketches(local, out)
    buf = torch.arange(0, 4096, dtype=torch.int32, device=dev)
    keep = buf.clone()
    c.allgatherv(buf, [64, 4096 * 4 - 128])
    rng = np.random.default_rng(4)
    S = _families(rng, 10, 10, 128)
    Y = torch.from_numpy(S.view(np.int32)).to(dev)
    N, s = Y.shape
    work = torch.zeros(mash.shared_counts_workspace_bytes(N, s, N, s), dtype=torch.uint8, device=dev)
    mash.index_build_part_dev(Y, 0, 1, work)
    c.index_allgather(N, s, work)
    counts = torch.zeros((N, N), dtype=torch.int16, device=dev)
    mash.shared_counts_reuse_dev(Y, Y, counts, work)
    torch.cuda.synchronize()
    assert torch.equal(out, local) and torch.equal(buf, keep)
    got = counts.cpu().numpy().view(np.uint16)
    for i in (0, 37, 99):
        for j in range(N):
            assert got[i, j] == orc.mash_shared(S[i], S[j])
    c.close()


def _gpus():
    import torch
    return torch.cuda.device_count()


def test_comm_multiprocess_python(tmp_path):
    """N processes, one per GPU, talking through libpolyhip's own communicator (no torch.distributed)"""
    n = min(_gpus(), 4)
    if n < 2:
        pytest.skip("needs >= 2 GPUs: RCCL refuses two ranks on one device")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    idfile = str(tmp_path / "uid.bin")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "comm_worker.py"), str(r), str(n), idfile,
                               str(tmp_path)], env=env) for r in range(n)]
    rcs = [p.wait(timeout=600) for p in procs]
    assert rcs == [0] * n
    import importlib.util
    spec = importlib.util.spec_from_file_location("comm_worker", os.path.join(ROOT, "tests", "comm_worker.py"))
    cw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cw)
    rank_sketches = cw.rank_sketches
    S = np.concatenate([rank_sketches(r, 12, 10, 256) for r in range(n)])
    for r in range(n):
        rep = json.load(open(tmp_path / f"rank_{r}.json"))
        assert rep["gather_ok"] and rep["counts_equal_local_index"] and rep["nranks"] == n
        got = np.load(tmp_path / f"counts_{r}.npy")
        for i in (0, 57, 119):
            for j in range(0, S.shape[0], 3):
                assert got[i, j] == orc.mash_shared(S[r * 120 + i], S[j])


def _abi_allgather_exe():
    from poly_amd import build
    return build.build_abi_allgather()


def test_abi_allgather_torch_free_one_rank():
    """tests/abi/abi_allgather.c: a plain C host (libpolyhip + the HIP runtime, no Python in the ranks) forks its ranks,
    passes the id through pipes and runs all-gather + index parts + row block; 1 rank on a 1-GPU box"""
    exe = _abi_allgather_exe()
    r = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_allgather ok: 1 rank" in r.stdout


def test_abi_allgather_torch_free_multi_rank():
    n = min(_gpus(), 4)
    if n < 2:
        pytest.skip("needs >= 2 GPUs: RCCL refuses two ranks on one device")
    exe = _abi_allgather_exe()
    r = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"abi_allgather ok: {n} rank" in r.stdout
