#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_distance_gpu.py tests/test_comm_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python scripts/fuzz_r03.py 60 77 2>&1 | tail -2
python scripts/quick_k2_r03.py full 2>&1 | tail -4
NSHOW=5 bash scripts/gpurun_r03/r03_k2idx.sh "$@"
