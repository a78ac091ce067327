#!/bin/bash
# round 5, call 21: where a row of the dense join goes -- ablation builds (timing only)
mkdir -p gpurun_out
( for v in "" j_F_NOSTORE j_F_NOLDS j_NOWALK j_NOWALK_NOSTORE; do
  if [ -z "$v" ]; then echo -n "product: "; timeout 120 python scripts/quick_k2_join_time.py 2>&1 | grep join
  else echo -n "$v: "; POLYHIP_LIB=poly_amd/libpolyhip_$v.so timeout 120 python scripts/quick_k2_join_time.py 2>&1 | grep join; fi
done ) > gpurun_out/c22_flush_ablation.log 2>&1
cat gpurun_out/c22_flush_ablation.log
