#!/bin/bash
# dense join with neighbouring columns in neighbouring dwords: timing + K2 tests
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/quick_k2d.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r02_k2j.log
timeout 900 python -m pytest tests/test_distance_gpu.py tests/test_stress_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 | tee gpurun_out/r02_k2j_tests.log
