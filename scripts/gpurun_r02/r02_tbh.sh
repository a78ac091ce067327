#!/bin/bash
# half-float two-band traceback: traceback / align tests, config-4 timing both ways
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_traceback_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|assert|^E " | tail -12 | tee gpurun_out/r02_tbh_tests.log
for h in 1 0; do
POLYHIP_TB_F16=$h python - <<'P' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02_tbh.log
import os, sys, torch
sys.path.insert(0,'.')
from poly_amd import bench_extra, align
r = bench_extra.sw(torch.device('cuda:0'))
print('TB_F16=' + os.environ['POLYHIP_TB_F16'], align.sw_traceback_last_half(), {k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ('score_pass_ms','traceback_ms','align_one_call_ms','mean_score','mean_alignment_len')})
P
done
