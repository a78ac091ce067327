#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_primers_gpu.py tests/test_pcr_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
python - <<'P' 2>&1 | grep -v amdgpu.ids
import sys, torch, json
sys.path.insert(0,'.')
from poly_amd import bench_extra
r = bench_extra.e2e(torch.device('cuda:0'))
for k,v in r.items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!='workload'})
print(bench_extra.tm_scan(torch.device('cuda:0')))
P
