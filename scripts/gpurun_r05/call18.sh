#!/bin/bash
# round 5, call 18: zero-ahead join -- where the time went (ablation builds)
mkdir -p gpurun_out
( for v in "" za_NOSTORE za_PLAINSTORE; do
  if [ -z "$v" ]; then echo -n "product: "; timeout 120 python scripts/quick_k2_join_time.py 2>&1 | grep join
       echo -n "product, whole-row flush: "; POLYHIP_K2_ZAHEAD=0 timeout 120 python scripts/quick_k2_join_time.py 2>&1 | grep join
  else echo -n "$v: "; POLYHIP_LIB=poly_amd/libpolyhip_$v.so timeout 120 python scripts/quick_k2_join_time.py 2>&1 | grep join; fi
done ) > gpurun_out/c18_za_ablation.log 2>&1
cat gpurun_out/c18_za_ablation.log
