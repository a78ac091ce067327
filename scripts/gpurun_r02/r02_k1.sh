#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/quick_k1_ab.py 200000 2>&1 | tee gpurun_out/r02_k1_ab.log
timeout 900 python -m pytest tests/test_mash_gpu.py tests/test_stress_gpu.py -x -q 2>&1 | tail -5 | tee gpurun_out/r02_k1_tests.log
timeout 200 python scripts/fuzz_k1.py 90 4242 2>&1 | tail -4 | tee gpurun_out/r02_k1_fuzz.log
POLYHIP_K1_SLABS=0 timeout 100 python scripts/fuzz_k1.py 30 77 2>&1 | tail -2
python scripts/quick_k1_lowc.py 2>&1 | tail -8 | tee gpurun_out/r02_k1_lowc.log
