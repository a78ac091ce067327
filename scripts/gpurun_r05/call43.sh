#!/bin/bash
# round 5, call 43: kernel statistics of the whole bench and of the 1 kb SW leg (with counters) on the final build
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
R=r05; ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; out=$ROOT/gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
  ( cd $ROOT && timeout 400 rocprofv3 "$@" ) > $out/run.log 2>&1
  f=$(find $out -name "*results.db" | head -1)
  if [ -n "$f" ]; then ( cd $ROOT && python scripts/rocpd_summary.py $f "$tag" > gpurun_out/$tag.md 2>&1 ); else echo "no db for $tag"; tail -5 $out/run.log; fi
  rm -rf $out; }
run ${R}_sw1kb_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_sw1kb_stats -o x -- python scripts/quick_sw_1kb_prof.py
run ${R}_sw1kb_pmc --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $ROOT/gpurun_out/prof_${R}_sw1kb_pmc -o x -- python scripts/quick_sw_1kb_prof.py
run ${R}_bench_full_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_bench_full_stats -o x -- python bench.py --no-cpu-baseline
cd $ROOT; grep -E "polyhip" gpurun_out/${R}_sw1kb_stats.md | head -4 | cut -c1-160; grep -E "sw_pkb" gpurun_out/${R}_sw1kb_pmc.md | grep INSTS_VALU | cut -c1-170; grep -E "tb_wave|sw_wave8|sw_pkb" gpurun_out/${R}_bench_full_stats.md | head -5 | cut -c1-150
