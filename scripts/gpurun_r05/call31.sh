#!/bin/bash
# round 5, call 31: the one-wave-per-pair traceback on a byte profile of the pair (path 7): parity tests of the long-read
# paths, the two forms side by side at bench sizes, a short fuzz
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_traceback_gpu.py -x -q -m gpu -k "long_reads" 2>&1 | tail -5
  timeout 300 python scripts/quick_tb_wave8.py
  timeout 400 python scripts/fuzz_k3.py 90 5000 2>&1 | tail -4 ) 2>&1 | tee gpurun_out/c31_tb_wave8.log
