"""K2 index at config 3: the default build against the build by eighths of the value range (POLYHIP_K2_SLICED=1); counts compared."""
import os
import sys
import torch
sys.path.insert(0, '.')
from poly_amd import bench_extra, mash
from poly_amd.bench_extra import _time
dev = torch.device('cuda:0')
s = 1000
sk = bench_extra.family_sketches(dev, 1000, 100, 10_000, 21, s, 0xC3)
N = sk.shape[0]
nrows = N // 8
X = sk[:nrows]
counts = torch.full((nrows, N), -1, dtype=torch.int16, device=dev)
os.environ["POLYHIP_K2_SLICED"] = "1"
work = torch.empty(mash.shared_counts_workspace_bytes(nrows, s, N, s), dtype=torch.uint8, device=dev)
res = {}
for tag, env in (("default", None), ("sliced", "1")):
    if env is None:
        os.environ.pop("POLYHIP_K2_SLICED", None)
    else:
        os.environ["POLYHIP_K2_SLICED"] = env
    ms_index = _time(lambda: mash.index_build_dev(sk, work), 10)
    ms_one = _time(lambda: mash.shared_counts_dev(X, sk, counts, work), 10)
    torch.cuda.synchronize()
    res[tag] = counts.clone()
    print(f"{tag}: item bytes {mash.index_item_bytes(work)}  index {ms_index:.3f} ms  one-shot {ms_one:.3f} ms  nonzero {int((counts != 0).sum())}", flush=True)
print("counts equal:", bool(torch.equal(res["default"], res["sliced"])))
