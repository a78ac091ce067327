#!/bin/bash
# round 5, call 17: the dense join with the next row's zeros written during the walk (zero-ahead)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_distance_gpu.py tests/test_index_build_gpu.py -x -q 2>&1 | tail -6 > gpurun_out/c17_tests.log; cat gpurun_out/c17_tests.log
timeout 300 python scripts/quick_k2_b4.py > gpurun_out/c17_k2.log 2>&1; cat gpurun_out/c17_k2.log
timeout 300 python scripts/quick_k2_full.py 2>&1 | tail -4
