#!/bin/bash
# round 6, verdict item 5: parity depth -- sweep_long2 (every pair of four legs vs the oracle), the extended tests
mkdir -p gpurun_out
timeout 1500 python scripts/sweep_long2.py 5000 > gpurun_out/r06_sweep_long2.log 2>&1; echo "sweep_long2 rc=$?"; cat gpurun_out/r06_sweep_long2.log
timeout 1500 python -m pytest tests/test_traceback_gpu.py -x -q -k "long_reads" 2>&1 | tail -4 | tee gpurun_out/r06_tests_long_reads.log
timeout 900 python -m pytest tests/test_distance_gpu.py -x -q -k "full_size" 2>&1 | tail -4 | tee gpurun_out/r06_tests_k2_full.log
timeout 900 python -m pytest tests/test_seqhash_gpu.py tests/test_clone_gpu.py -x -q 2>&1 | tail -3 | tee gpurun_out/r06_tests_seqhash.log
