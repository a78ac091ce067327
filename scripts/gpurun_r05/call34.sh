#!/bin/bash
# round 5, call 34: the one-wave-per-pair traceback: byte-profile sweep + scalar walk (the byte-profile kernel's score from its
# plane); the bit push through v_sub + v_alignbit as a variant build; parity tests, the forms side by side, a short fuzz
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_traceback_gpu.py tests/test_align_gpu.py -x -q -m gpu 2>&1 | tail -5
  timeout 300 python scripts/quick_tb_wave8.py
  echo "== variant: v_sub + v_alignbit"
  POLYHIP_LIB=poly_amd/libpolyhip_tbalign.so timeout 300 python scripts/quick_tb_wave8.py
  timeout 400 python scripts/fuzz_k3.py 90 11000 2>&1 | tail -3 ) 2>&1 | tee gpurun_out/c34_tb_wave8.log
