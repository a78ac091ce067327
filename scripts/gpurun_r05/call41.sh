#!/bin/bash
# round 5, call 41: the int16 packed score kernels' block maximum through v_pk_maximum3_f16 (integers below 0x7C00 order like
# halves): parity tests of the score paths, the long-read legs, a short fuzz
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_align_gpu.py -x -q -m gpu 2>&1 | tail -3
  timeout 300 python scripts/quick_sw_long.py
  timeout 300 python scripts/fuzz_k3.py 60 21000 2>&1 | tail -2 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c41_pk_max3.log
