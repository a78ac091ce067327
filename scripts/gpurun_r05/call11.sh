#!/bin/bash
# round 5, call 11: the exhaustive full-size parity tests (configs 2, 4, 5), the 100k-pair sweep, K2 profiles after maxlast
mkdir -p gpurun_out
nproc > gpurun_out/c11_nproc.log
( time timeout 900 python -m pytest tests/test_mash_gpu.py -x -q -k "full_size_config2" ) 2>&1 | tail -8 > gpurun_out/c11_cfg2.log; cat gpurun_out/c11_cfg2.log
( time timeout 900 python -m pytest tests/test_primers_gpu.py -x -q -k "full_size_config5" ) 2>&1 | tail -8 > gpurun_out/c11_cfg5.log; cat gpurun_out/c11_cfg5.log
( time timeout 900 python -m pytest tests/test_traceback_gpu.py -x -q -k "config4_full_size" ) 2>&1 | tail -8 > gpurun_out/c11_cfg4.log; cat gpurun_out/c11_cfg4.log
( time timeout 900 python scripts/sweep_full.py 100000 ) > gpurun_out/r05_sweep_full.log 2>&1; tail -6 gpurun_out/r05_sweep_full.log
timeout 300 python scripts/quick_k2_b4.py > gpurun_out/c11_k2_b4.log 2>&1; cat gpurun_out/c11_k2_b4.log
