#!/bin/bash
R=r03
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { # tag, rocprof args..., -- cmd
  tag=$1; shift
  out=$ROOT/gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
  ( cd $ROOT && rocprofv3 "$@" ) > $out/run.log 2>&1
  f=$(find $out -name "*results.db" | head -1)
  if [ -n "$f" ]; then ( cd $ROOT && python scripts/rocpd_summary.py $f "$tag" > gpurun_out/$tag.md 2>&1 ); else echo "no db for $tag"; tail -5 $out/run.log; fi
  rm -rf $out
}
run ${R}_k2_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_k2_stats -o x -- python scripts/quick_k2c.py
run ${R}_k2_pmc_sq --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $ROOT/gpurun_out/prof_${R}_k2_pmc_sq -o x -- python scripts/quick_k2c.py
grep polyhip $ROOT/gpurun_out/${R}_k2_stats.md | head -14 | cut -c1-150
grep "rowjoin_dense\|fine_kernel\|coarse_scatter_staged\|check_kernel" $ROOT/gpurun_out/${R}_k2_pmc_sq.md | grep -v "calls" | cut -c1-160 | head -40
