#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_distance_gpu.py tests/test_comm_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r03_k2_tests.log
cat gpurun_out/r03_k2_tests.log
python scripts/quick_k2_r03.py full 2>&1 | tail -8
