#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "" _k5t128 _k5t64; do
  if [ -z "$v" ]; then unset POLYHIP_LIB; else export POLYHIP_LIB=$PWD/poly_amd/libpolyhip$v.so; fi
  echo "== variant '$v'"; python scripts/quick_k5.py 2>&1 | grep -E "random|unit 8|poly-A"
done
