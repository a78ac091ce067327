#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in A B C D A B; do ./scripts/ubench/k1_w_$t; done 2>&1 | tee gpurun_out/r02_k1_variants.log
POLYHIP_K1_SLABS=0 ./scripts/ubench/k1_w_A 2>&1 | tee -a gpurun_out/r02_k1_variants.log
python scripts/quick_k1_ab.py 200000 2>&1 | tee gpurun_out/r02_k1_ab2.log
timeout 900 python -m pytest tests/test_mash_gpu.py tests/test_stress_gpu.py -x -q 2>&1 | tail -3
timeout 100 python scripts/fuzz_k1.py 40 999 2>&1 | tail -2
