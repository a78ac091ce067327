#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_distance_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5 | tee gpurun_out/r02_k2_tests.log
timeout 600 python scripts/quick_k2c.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_k2_abl.log
