#!/bin/bash
# K2 join walk variants (round 6): two groups of buckets in flight per wave, group sizes 4 / 8 / 16
mkdir -p gpurun_out
{
for i in 1 2; do
for v in "" k2pipe8 k2pipe4 k2u4 k2u16 k2pipe16; do
  if [ -z "$v" ]; then echo -n "default: "; python scripts/quick_k2_join_time.py 2>&1 | tail -1
  else echo -n "$v: "; POLYHIP_LIB=poly_amd/libpolyhip_$v.so python scripts/quick_k2_join_time.py 2>&1 | tail -1; fi
done
done
} > gpurun_out/r06_k2_join_pipe.log 2>&1
cat gpurun_out/r06_k2_join_pipe.log
