#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "" poly_amd/libpolyhip_g4.so poly_amd/libpolyhip_g8.so poly_amd/libpolyhip_g32.so; do echo "== $v"; POLYHIP_LIB=$v python scripts/quick_feeders.py 2>&1 | grep -v amdgpu.ids | tail -1; done
