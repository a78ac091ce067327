#!/bin/bash
# round 5, call 36: long reads, score + strings in one call: the end cell of a maximum in one block left to the byte-profile
# one-wave-per-pair traceback; parity tests, both forms side by side, a short fuzz
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_traceback_gpu.py tests/test_align_gpu.py -x -q -m gpu 2>&1 | tail -5
  timeout 300 python scripts/quick_sw_onecall.py
  timeout 300 python scripts/quick_tb_wave8.py 2>&1 | grep " x "
  timeout 400 python scripts/fuzz_k3.py 60 15000 2>&1 | tail -2 ) 2>&1 | tee gpurun_out/c36_sw_onecall.log
