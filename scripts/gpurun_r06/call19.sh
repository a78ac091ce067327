#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_seqhash_gpu.py tests/test_clone_gpu.py -x -q 2>&1 | grep -E "passed|failed"
{
echo "== global loads (default)"; python scripts/quick_seqhash.py
echo "== flat loads (PH_S2_GLOBAL_LOADS=0)"; POLYHIP_LIB=poly_amd/libpolyhip_s2flat.so python scripts/quick_seqhash.py
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_seqhash_global_loads.log
