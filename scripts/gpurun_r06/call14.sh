#!/bin/bash
# K2 join probes: ONE aligned 16-byte (8-byte) load per lane and bucket instead of two dword loads
mkdir -p gpurun_out
{
for i in 1 2; do
for v in "" k2x4a k2x2a k2x4a_u4; do
  if [ -z "$v" ]; then echo -n "default: "; python scripts/quick_k2_join_time.py 2>&1 | tail -1
  else echo -n "$v: "; POLYHIP_LIB=poly_amd/libpolyhip_$v.so python scripts/quick_k2_join_time.py 2>&1 | tail -1; fi
done
done
} > gpurun_out/r06_k2_join_wide.log 2>&1
cat gpurun_out/r06_k2_join_wide.log
