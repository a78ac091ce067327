#!/bin/bash
# round 5, call 33: the one-wave-per-pair traceback: byte-profile sweep + the walk out of the wave's registers; parity tests of
# every test that reaches a wave kernel, the forms side by side at bench sizes, a short fuzz
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_traceback_gpu.py tests/test_align_gpu.py -x -q -m gpu 2>&1 | tail -5
  timeout 300 python scripts/quick_tb_wave8.py
  timeout 400 python scripts/fuzz_k3.py 90 9000 2>&1 | tail -3 ) 2>&1 | tee gpurun_out/c33_tb_wave8.log
