#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/quick_k1_lowc.py 2>&1 | grep -v amdgpu.ids | tail -5
python scripts/quick_k1_ab.py 200000 2>&1 | grep -v amdgpu.ids | head -2
timeout 900 python -m pytest tests/test_mash_gpu.py tests/test_stress_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
