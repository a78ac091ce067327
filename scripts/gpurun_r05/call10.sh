#!/bin/bash
# round 5, call 10: every K2-related test file with the new default build, then the distance leg of the bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_index_build_gpu.py tests/test_distance_gpu.py tests/test_multidev_gpu.py tests/test_comm_gpu.py tests/test_abi_gpu.py -x -q 2>&1 | tail -6 > gpurun_out/c10_tests.log
cat gpurun_out/c10_tests.log
timeout 300 python - > gpurun_out/c10_dist_leg.log 2>&1 <<'PY'
import json, sys, torch
sys.path.insert(0, '.')
from poly_amd import bench_extra
r = bench_extra.mash_distance(torch.device('cuda:0'))
r.pop('_spot', None)
print(json.dumps(r, indent=1))
PY
cat gpurun_out/c10_dist_leg.log | head -60
