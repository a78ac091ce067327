#!/bin/bash
# half-float row without the diagonal copy: K3 tests, config-4 and 250-bp timing, fuzz
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_traceback_gpu.py tests/test_align_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -8 | tee gpurun_out/r02_k3h3_tests.log
python - <<'P' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02_k3h3.log
import os, sys, torch
sys.path.insert(0,'.')
from poly_amd import bench_extra
for args in ((), (400_000, 250)):
    r = bench_extra.sw(torch.device('cuda:0'), *args)
    print(args, {k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ('score_pass_ms','traceback_ms','align_one_call_ms','cell_updates_per_s','frac_of_valu_issue_ceiling','mean_score')})
P
timeout 300 python scripts/fuzz_k3.py 120 2>&1 | tail -2
