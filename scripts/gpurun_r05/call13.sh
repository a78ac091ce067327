#!/bin/bash
# round 5, call 13: multi-device tests after the hardening (own-rows storage, info struct), abi tests, seqhash/clone (AuxStream/SyncOnExit)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_multidev_gpu.py tests/test_abi_gpu.py tests/test_comm_gpu.py tests/test_seqhash_gpu.py tests/test_clone_gpu.py tests/test_distance_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/c13_tests.log
cat gpurun_out/c13_tests.log
