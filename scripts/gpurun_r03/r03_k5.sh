#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_seqhash_gpu.py tests/test_clone_gpu.py -x -q -m gpu 2>&1 | tail -5 | grep -E "passed|failed|Error"
python scripts/quick_k5.py 2>&1 | grep -v amdgpu
