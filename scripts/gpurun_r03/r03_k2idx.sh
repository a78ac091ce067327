#!/bin/bash
# usage: r03_k2idx.sh [variant tag ...]: per-kernel times of the index build, product library first, then variants
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "" "$@"; do
  if [ -z "$v" ]; then unset POLYHIP_LIB; else export POLYHIP_LIB=$ROOT/poly_amd/libpolyhip_$v.so; fi
  tag=r03_k2idx_${v:-product}
  out=$ROOT/gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
  ( cd $ROOT && rocprofv3 --kernel-trace --stats -d $out -o x -- python scripts/quick_k2_index.py ) > $out/run.log 2>&1
  f=$(find $out -name "*results.db" | head -1)
  if [ -n "$f" ]; then ( cd $ROOT && python scripts/rocpd_summary.py $f "$tag" > gpurun_out/$tag.md 2>&1 ); else echo "no db for $tag"; tail -5 $out/run.log; fi
  rm -rf $out
  echo "== ${v:-product}"; grep "polyhip::k2" $ROOT/gpurun_out/$tag.md | head -${NSHOW:-4} | cut -c1-120
done
