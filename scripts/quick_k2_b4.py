"""K2 index at config 3 (100,000 sketches of 1000): the two-level build (POLYHIP_K2_B4=0) against the sliced build on 4-byte
intermediate items (round 5), with the sliced build's two stage shapes and a few slice lengths; counts compared."""
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
work = torch.empty(mash.shared_counts_workspace_bytes(nrows, s, N, s), dtype=torch.uint8, device=dev)
KEYS = ("POLYHIP_K2_B4", "POLYHIP_K2_B4_SLOTS", "POLYHIP_K2_B4_TL", "POLYHIP_K2_ZAHEAD")
variants = [("two-level", {"POLYHIP_K2_B4": "0"}), ("b4 default", {}), ("b4, zero-ahead join", {"POLYHIP_K2_ZAHEAD": "1"})]
if len(sys.argv) > 1 and sys.argv[1] == "sweep":
    variants += [(f"b4 slots128 tl{t}", {"POLYHIP_K2_B4_SLOTS": "128", "POLYHIP_K2_B4_TL": str(t)}) for t in (72, 83, 92)]
    variants += [(f"b4 slots64 tl{t}", {"POLYHIP_K2_B4_SLOTS": "64", "POLYHIP_K2_B4_TL": str(t)}) for t in (36, 41, 46)]
    variants += [(f"b4 slots32 tl{t}", {"POLYHIP_K2_B4_SLOTS": "32", "POLYHIP_K2_B4_TL": str(t)}) for t in (16, 18, 20, 23)]
ref = None
for tag, env in variants:
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    ms_index = _time(lambda: mash.index_build_dev(sk, work), 10)
    ms_one = _time(lambda: mash.shared_counts_dev(X, sk, counts, work), 10)
    ms_join = _time(lambda: mash.shared_counts_reuse_dev(X, sk, counts, work), 10)
    torch.cuda.synchronize()
    info = mash.index_build_info(work)
    same = ""
    if ref is None:
        ref = counts.clone()
    else:
        same = f"  counts equal: {bool(torch.equal(ref, counts))}"
    print(f"{tag}: item bytes {mash.index_item_bytes(work)}  index {ms_index:.3f} ms  one-shot {ms_one:.3f} ms  join {ms_join:.3f} ms  nonzero {int((counts != 0).sum())}  {info}{same}", flush=True)
for k in KEYS:
    os.environ.pop(k, None)
