#!/bin/bash
# round 5, call 25: the join's flush with ordinary stores instead of nontemporal ones
( echo -n "product: "; timeout 120 python scripts/quick_k2_join_time.py 2>&1 | grep join
  echo -n "j_F_PLAIN: "; POLYHIP_LIB=poly_amd/libpolyhip_j_F_PLAIN.so timeout 120 python scripts/quick_k2_join_time.py 2>&1 | grep join
  echo -n "product: "; timeout 120 python scripts/quick_k2_join_time.py 2>&1 | grep join ) 2>&1 | tee gpurun_out/c25_plain.log
