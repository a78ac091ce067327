#!/bin/bash
# Round 6 rocprofv3 evidence (run through gpurun): summaries land in gpurun_out/r05_*.md, copy them into profiles/.
# Usage: collect_profiles_r06.sh [set ...]   sets: bench k1 k2 k3 legs   (default: all)
R=r06
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
SETS="${@:-bench k1 k2 k3 legs json}"
for s in $SETS; do case $s in
bench)
  run ${R}_bench_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_bench_stats -o x -- python bench.py --no-extra --no-cpu-baseline
  run ${R}_bench_full_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_bench_full_stats -o x -- python bench.py --no-cpu-baseline
  run ${R}_bench_fetch --pmc FETCH_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_bench_fetch -o x -- python bench.py --no-extra --no-cpu-baseline
  run ${R}_bench_write --pmc WRITE_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_bench_write -o x -- python bench.py --no-extra --no-cpu-baseline
  ;;
k1)
  run ${R}_k1_issue --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $ROOT/gpurun_out/prof_${R}_k1_issue -o x -- python scripts/quick_k1.py
  ;;
k2)
  run ${R}_k2_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_k2_stats -o x -- python scripts/quick_k2c.py
  run ${R}_k2_pmc_sq --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $ROOT/gpurun_out/prof_${R}_k2_pmc_sq -o x -- python scripts/quick_k2c.py
  run ${R}_k2_fetch --pmc FETCH_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_k2_fetch -o x -- python scripts/quick_k2c.py
  run ${R}_k2_write --pmc WRITE_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_k2_write -o x -- python scripts/quick_k2c.py
  ;;
k2stats)
  run ${R}_k2_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_k2_stats -o x -- python scripts/quick_k2c.py
  ;;
k2traffic)
  run ${R}_k2_fetch --pmc FETCH_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_k2_fetch -o x -- python scripts/quick_k2c.py
  run ${R}_k2_write --pmc WRITE_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_k2_write -o x -- python scripts/quick_k2c.py
  ;;
k3)
  POLYHIP_SW_OVERLAP=0 run ${R}_k3_stats_nooverlap --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_k3_stats_nooverlap -o x -- python scripts/quick_k3tb.py
  POLYHIP_SW_OVERLAP=0 run ${R}_k3_pmc_nooverlap --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $ROOT/gpurun_out/prof_${R}_k3_pmc_nooverlap -o x -- python scripts/quick_k3tb.py
  ;;
legs)
  for g in A B; do
    run ${R}_legs${g}_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_legs${g}_stats -o x -- python scripts/quick_legs.py $g
    run ${R}_legs${g}_fetch --pmc FETCH_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_legs${g}_fetch -o x -- python scripts/quick_legs.py $g
    run ${R}_legs${g}_write --pmc WRITE_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_legs${g}_write -o x -- python scripts/quick_legs.py $g
  done
  ;;
json)
  ( cd $ROOT && python scripts/traffic_json.py $R gpurun_out > gpurun_out/traffic.json; python scripts/issue_json.py $R gpurun_out > gpurun_out/issue_json.log 2>&1 )
  ;;
esac; done
for t in $(cd $ROOT/gpurun_out && ls ${R}_*.md 2>/dev/null); do echo "== $t"; grep -E "polyhip" $ROOT/gpurun_out/$t | head -8 | cut -c1-170; done
