#!/bin/bash
# round 5, call 7: where fine4's time goes -- ablation builds (timing only, their indexes are wrong)
mkdir -p gpurun_out
( for v in "" f4_NORANK f4_RANKED; do
  if [ -z "$v" ]; then echo -n "product: "; timeout 120 python scripts/quick_k2_index_time.py 2>&1 | grep index
  else echo -n "$v: "; POLYHIP_LIB=poly_amd/libpolyhip_$v.so timeout 120 python scripts/quick_k2_index_time.py 2>&1 | grep index; fi
done ) > gpurun_out/c08_f4_ablation.log 2>&1
cat gpurun_out/c08_f4_ablation.log
