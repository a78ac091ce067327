#!/bin/bash
# round 5, call 45: the byte-profile traceback without / with the locate question as two instantiations: tests, the 1 kb leg's times
mkdir -p gpurun_out
( time timeout 600 python -m pytest tests/test_traceback_gpu.py -m gpu -x -q ) > gpurun_out/c45_traceback_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/c45_traceback_tests.log | tail -3
timeout 200 python scripts/quick_sw_onecall.py 2>&1 | grep " x "
timeout 200 python scripts/quick_tb_wave8.py 2>&1 | grep " x " | head -2
