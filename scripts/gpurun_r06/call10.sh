#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_seqhash_gpu.py tests/test_clone_gpu.py -x -q 2>&1 | tail -3
{
echo "== two blocks per round of loads (default)"; python scripts/quick_seqhash.py
echo "== one block per round (PH_S2_PAIR_BLOCKS=0)"; POLYHIP_LIB=poly_amd/libpolyhip_s2nopair.so python scripts/quick_seqhash.py
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_seqhash_pair_blocks.log
