#!/bin/bash
# verdict r02 item 2: K3 with POLYHIP_SW_OVERLAP=0 / POLYHIP_TB_OVERLAP=0 -- kernel durations that SUM to the pass
R=r03
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { # tag, rocprof args..., -- cmd
  tag=$1; shift
  out=$ROOT/gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
  ( cd $ROOT && rocprofv3 "$@" ) > $out/run.log 2>&1
  f=$(find $out -name "*results.db" | head -1)
  if [ -n "$f" ]; then ( cd $ROOT && python scripts/rocpd_summary.py $f "$tag" > gpurun_out/$tag.md 2>&1 ); else echo "no db for $tag"; tail -5 $out/run.log; fi
  grep "K3 score" $out/run.log >> $ROOT/gpurun_out/${R}_k3_lines.log
  rm -rf $out
}
export POLYHIP_SW_OVERLAP=0 POLYHIP_TB_OVERLAP=0
( cd $ROOT && python scripts/quick_k3tb.py ) | grep "K3 score" > $ROOT/gpurun_out/${R}_k3_lines.log
run ${R}_k3_stats_nooverlap --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_k3_stats_nooverlap -o x -- python scripts/quick_k3tb.py
run ${R}_k3_pmc_nooverlap --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d $ROOT/gpurun_out/prof_${R}_k3_pmc_nooverlap -o x -- python scripts/quick_k3tb.py
unset POLYHIP_SW_OVERLAP POLYHIP_TB_OVERLAP
( cd $ROOT && python scripts/quick_k3tb.py ) | grep "K3 score" >> $ROOT/gpurun_out/${R}_k3_lines.log
cat $ROOT/gpurun_out/${R}_k3_lines.log
grep polyhip $ROOT/gpurun_out/${R}_k3_stats_nooverlap.md | head -12 | cut -c1-160
