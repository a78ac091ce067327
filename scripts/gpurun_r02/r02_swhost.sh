#!/bin/bash
# SW align host flavour: chunk pipeline test + wall time per chunk count
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_traceback_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 | tee gpurun_out/r02_swhost_tests.log
timeout 600 python scripts/quick_sw_host.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_swhost.log
