#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_distance_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
python scripts/quick_k2d.py 2>&1 | grep -v amdgpu.ids
python scripts/quick_k2b.py 2>&1 | grep -v amdgpu.ids
python scripts/quick_k2_dense.py 2>&1 | grep -v amdgpu.ids | tail -4
timeout 60 python scripts/fuzz_misc.py 40 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
