#!/bin/bash
# round 5, call 26: what check4 costs -- ablation builds under the profiler (the kernel's own time)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in "" c4_NOHIST c4_NOPOS c4_NEITHER; do
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_c4; rm -rf $out; mkdir -p $out
  ( cd $GRAFT_REPO_ROOT && POLYHIP_LIB=${v:+poly_amd/libpolyhip_$v.so} rocprofv3 --kernel-trace --stats -d $out -o x -- python scripts/quick_k2_index.py ) > $out/run.log 2>&1
  f=$(find $out -name "*results.db" | head -1)
  ( cd $GRAFT_REPO_ROOT && python scripts/rocpd_summary.py $f "c4 $v" 2>/dev/null | grep -E "check4|scatter4|fine4" | cut -c1-110 | sed "s/^/[${v:-product}] /" )
  rm -rf $out
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/c26_check4_ablation.log
