#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_seqhash_gpu.py tests/test_clone_gpu.py -x -q 2>&1 | tail -5 | tee gpurun_out/r06_tests_seqhash_fold.log
{
echo "== fold (default)"; python scripts/quick_seqhash.py
echo "== POLYHIP_S2_FOLD=0"; POLYHIP_S2_FOLD=0 python scripts/quick_seqhash.py
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_seqhash_fold.log
timeout 600 python scripts/fuzz_k5.py 2>&1 | tail -3 | tee gpurun_out/r06_fuzz_k5.log
