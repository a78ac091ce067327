#!/bin/bash
# round 5, call 24: seqhash with the streaming normalise pass
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_seqhash_gpu.py tests/test_clone_gpu.py tests/test_stress_gpu.py tests/test_multidev_gpu.py -x -q -k "seqhash or clone or hash or stress" 2>&1 | tail -4
POLYHIP_S2_STREAM=0 timeout 300 python -m pytest tests/test_seqhash_gpu.py -x -q 2>&1 | tail -2
timeout 200 python scripts/quick_seqhash.py 2>&1 | tail -2
timeout 600 python scripts/fuzz_misc.py 2>&1 | tail -2
