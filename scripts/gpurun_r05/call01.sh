#!/bin/bash
# round 5, call 1: the sliced index build (B4) -- its own tests, the K2 tests five ways, timing sweep at config 3
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_index_build_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/c01_tests_b4.log
timeout 600 python -m pytest tests/test_distance_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/c01_tests_dist.log
timeout 300 python scripts/quick_k2_b4.py sweep > gpurun_out/c01_k2_b4.log 2>&1
tail -5 gpurun_out/c01_tests_b4.log; tail -5 gpurun_out/c01_tests_dist.log; cat gpurun_out/c01_k2_b4.log
