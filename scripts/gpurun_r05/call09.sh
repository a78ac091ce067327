#!/bin/bash
# round 5, call 7: where fine4's time goes -- ablation builds (timing only, their indexes are wrong)
mkdir -p gpurun_out
( for v in "" f4_P6 f4_P4 f4_P2 f4_P1; do
  if [ -z "$v" ]; then echo -n "product: "; timeout 120 python scripts/quick_k2_index_time.py 2>&1 | grep index
  else echo -n "$v: "; POLYHIP_LIB=poly_amd/libpolyhip_$v.so timeout 120 python scripts/quick_k2_index_time.py 2>&1 | grep index; fi
done ) > gpurun_out/c09_f4_pieces.log 2>&1
cat gpurun_out/c09_f4_pieces.log
timeout 600 python -m pytest tests/test_index_build_gpu.py -x -q 2>&1 | tail -3
