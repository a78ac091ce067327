import sys, torch
sys.path.insert(0, '.')
from poly_amd import bench_extra
dev = torch.device('cuda:0')
for n in (20_000, 80_000, 200_000):
    r = bench_extra.sw(dev, n, 1000)
    print(n, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in ('score_pass_ms', 'traceback_ms', 'cell_updates_per_s', 'score_path', 'traceback_path')}, flush=True)
