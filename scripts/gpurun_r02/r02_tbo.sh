#!/bin/bash
# traceback chunks overlapped on two streams: test + config-4 timing both ways
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_traceback_gpu.py -x -q -k "chunks_overlapped or config4_full or fused" 2>&1 | grep -E "passed|failed|Error|assert|^E " | tail -8 | tee gpurun_out/r02_tbo_tests.log
for o in 1 0; do
POLYHIP_TB_OVERLAP=$o python - <<'P' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02_tbo.log
import os, sys, torch
sys.path.insert(0,'.')
from poly_amd import bench_extra
r = bench_extra.sw(torch.device('cuda:0'))
print('OVERLAP=' + os.environ['POLYHIP_TB_OVERLAP'], {k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ('score_pass_ms','traceback_ms','align_one_call_ms')})
P
done
