"""configs[3] in ONE call (polyhip_sw_align_batch_dev: score + end cell + both aligned strings), ms per 1M pairs"""
import sys, torch
sys.path.insert(0, '.')
from poly_amd import bench_extra
dev = torch.device('cuda:0')
for _ in range(2):
    r = bench_extra.sw(dev)
    print(f"one call {r['align_one_call_ms']:.2f} ms = {r['cell_updates_per_s_align_one_call']:.3e}; score pass {r['score_pass_ms']:.2f}; traceback {r['traceback_ms']:.2f}", flush=True)
