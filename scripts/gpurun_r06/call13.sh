#!/bin/bash
# K5: fewer list entries per wave (LDS per workgroup 36.5 KB -> 24 KB: six workgroups per CU instead of four)
mkdir -p gpurun_out
{
for v in "" k5w512 k5w256 k5w128; do
  echo "== ${v:-default (1024)}"
  if [ -z "$v" ]; then python scripts/quick_k5.py 2>&1 | grep -v amdgpu; python scripts/quick_seqhash.py 2>&1 | tail -1
  else POLYHIP_LIB=poly_amd/libpolyhip_$v.so python scripts/quick_k5.py 2>&1 | grep -v amdgpu; POLYHIP_LIB=poly_amd/libpolyhip_$v.so python scripts/quick_seqhash.py 2>&1 | tail -1; fi
done
} > gpurun_out/r06_k5_wlist.log 2>&1
cat gpurun_out/r06_k5_wlist.log
