#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_traceback_gpu.py tests/test_align_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
python scripts/quick_k3tb.py 2>&1 | grep "K3 score"
python scripts/quick_k3tb.py 2>&1 | grep "K3 score"
timeout 300 python scripts/fuzz_k3.py 40 2>&1 | tail -2
