#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_mash_gpu.py tests/test_abi_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r03_k1range_tests.log
cat gpurun_out/r03_k1range_tests.log
