#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/quick_k1_ab.py 200000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_k1_ab3.log
timeout 900 python -m pytest tests/test_mash_gpu.py tests/test_stress_gpu.py tests/test_primers_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
timeout 100 python scripts/fuzz_k1.py 50 31337 2>&1 | tail -2
POLYHIP_K1_SLABS=0 timeout 100 python scripts/fuzz_k1.py 25 5151 2>&1 | tail -1
python scripts/quick_k1_lowc.py 2>&1 | grep -v amdgpu.ids | tail -5
