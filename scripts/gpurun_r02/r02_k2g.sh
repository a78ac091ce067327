#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "" poly_amd/libpolyhip_u8.so poly_amd/libpolyhip_u2.so; do POLYHIP_LIB=$v python scripts/quick_k2d.py 2>&1 | grep -v amdgpu.ids; done
timeout 900 python -m pytest tests/test_distance_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
