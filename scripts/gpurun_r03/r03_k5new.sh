#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_seqhash_gpu.py tests/test_abi_gpu.py tests/test_clone_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 300 python scripts/fuzz_k5.py 50 1 2>&1 | tail -3
python scripts/quick_k5.py 2>&1 | grep -v amdgpu.ids | tail -8
python scripts/quick_k5b.py 2>&1 | grep L=; echo "== WAVE_MAX=32768"; POLYHIP_K5_WAVE_MAX=32768 python scripts/quick_k5b.py 2>&1 | grep "L=20000"
