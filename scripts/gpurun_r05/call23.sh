#!/bin/bash
# round 5, call 23: the join's walk on 8-byte loads
mkdir -p gpurun_out
timeout 120 python scripts/quick_k2_join_time.py 2>&1 | grep join
timeout 900 python -m pytest tests/test_distance_gpu.py tests/test_index_build_gpu.py -x -q 2>&1 | tail -4
timeout 300 python scripts/quick_k2_b4.py 2>&1 | grep -v amdgpu
timeout 300 python scripts/quick_k2_full.py 2>&1 | tail -2
