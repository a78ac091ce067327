"""Exercises bench.py's multi-GPU distance leg (RCCL init, all_gather, K2 row block) with a 1-rank nccl group."""
import os, sys
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
import torch, torch.distributed as dist
import bench
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
# monkeypatch: make the gather path run its collective even for a single rank
from poly_amd import sharding
orig = sharding.gather_sketches
def forced(local, group=None):
    out = torch.empty_like(local)
    dist.all_gather_into_tensor(out, local.contiguous())
    assert torch.equal(out, local)
    return out, 0
sharding.gather_sketches = forced
r = bench.allgather_distance(torch.device("cuda", 0), 0, 1)
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k != "workload"})
dist.destroy_process_group()
