#!/bin/bash
# half-float cell in the banded kernel too: K3 tests, 250-bp timing both ways, fuzz
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_traceback_gpu.py tests/test_align_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -8 | tee gpurun_out/r02_k3h2_tests.log
for h in 1 0; do
POLYHIP_SW_F16=$h python - <<'P' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02_k3h2.log
import os, sys, torch
sys.path.insert(0,'.')
from poly_amd import bench_extra
r = bench_extra.sw(torch.device('cuda:0'), 400_000, 250)
print('F16=' + os.environ['POLYHIP_SW_F16'], {k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k not in ('workload','roofline')})
P
done
timeout 300 python scripts/fuzz_k3.py 120 2>&1 | tail -3
