#!/bin/bash
# packed score pass in sub-batches on two streams: tests + config-4 timing both ways
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_traceback_gpu.py tests/test_align_gpu.py -x -q -k "config4 or fused or packed_pass" 2>&1 | grep -E "passed|failed|Error|assert|^E " | tail -8 | tee gpurun_out/r02_swo_tests.log
for o in 1 0; do
POLYHIP_SW_OVERLAP=$o python - <<'P' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02_swo.log
import os, sys, torch
sys.path.insert(0,'.')
from poly_amd import bench_extra
r = bench_extra.sw(torch.device('cuda:0'))
print('SW_OVERLAP=' + os.environ['POLYHIP_SW_OVERLAP'], {k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ('score_pass_ms','traceback_ms','align_one_call_ms','mean_score')})
P
done
