#!/bin/bash
# round 5, call 35: long reads against one reference: the locate step of the score pass on a byte profile (sw_wave8_kernel);
# parity tests of the align paths, the two forms side by side, a short fuzz
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_align_gpu.py tests/test_traceback_gpu.py -x -q -m gpu 2>&1 | tail -5
  timeout 300 python scripts/quick_sw_long.py
  timeout 400 python scripts/fuzz_k3.py 90 13000 2>&1 | tail -3 ) 2>&1 | tee gpurun_out/c35_sw_wave8.log
