#!/bin/bash
# Round 3 rocprofv3 evidence (run through gpurun): summaries land in gpurun_out/r03_*.md, copy them into profiles/.
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
run ${R}_bench_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_bench_stats -o x -- python bench.py --no-extra --no-cpu-baseline
run ${R}_bench_full_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_bench_full_stats -o x -- python bench.py --no-cpu-baseline
run ${R}_bench_fetch --pmc FETCH_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_bench_fetch -o x -- python bench.py --no-extra --no-cpu-baseline
run ${R}_bench_write --pmc WRITE_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_bench_write -o x -- python bench.py --no-extra --no-cpu-baseline
run ${R}_k2_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_k2_stats -o x -- python scripts/quick_k2c.py
run ${R}_k2_pmc_sq --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $ROOT/gpurun_out/prof_${R}_k2_pmc_sq -o x -- python scripts/quick_k2c.py
run ${R}_k2_fetch --pmc FETCH_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_k2_fetch -o x -- python scripts/quick_k2c.py
run ${R}_k2_write --pmc WRITE_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_k2_write -o x -- python scripts/quick_k2c.py
run ${R}_feeders_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_feeders_stats -o x -- python scripts/quick_feeders.py
for t in bench_stats bench_fetch bench_write k2_stats k2_pmc_sq k2_fetch k2_write feeders_stats; do echo "== $t"; grep -E "polyhip" $ROOT/gpurun_out/${R}_$t.md | head -8 | cut -c1-170; done
