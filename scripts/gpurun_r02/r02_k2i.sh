#!/bin/bash
# dense join: bucket loads software-pipelined / more in flight (variants built by scripts/build_variant.sh)
cd $GRAFT_REPO_ROOT
for t in p0u4 p0u8 p1u2 p1u4 p1u8; do
POLYHIP_LIB=poly_amd/libpolyhip_$t.so timeout 300 python scripts/quick_k2d.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r02_k2i.log
done
