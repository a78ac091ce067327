#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "" _b64 _b16 _b8; do
  if [ -z "$v" ]; then unset POLYHIP_LIB; else export POLYHIP_LIB=$PWD/poly_amd/libpolyhip$v.so; fi
  echo "== variant '$v'"; python scripts/quick_k2_r03.py full 2>&1 | grep -E "^compact|counts equal|^full"
done
