#!/bin/bash
# Round 2 rocprofv3 evidence (run through gpurun): summaries land in gpurun_out/r02_*.md, copy them into profiles/.
R=r02
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
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
run ${R}_bench_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_bench_stats -o x -- python bench.py --no-extra --no-cpu-baseline
run ${R}_bench_full_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_bench_full_stats -o x -- python bench.py --no-cpu-baseline
run ${R}_bench_fetch --pmc FETCH_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_bench_fetch -o x -- $BENCH
run ${R}_bench_write --pmc WRITE_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_bench_write -o x -- $BENCH
run ${R}_k1_issue --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace -d $ROOT/gpurun_out/prof_${R}_k1_issue -o x -- python scripts/quick_k1.py
run ${R}_k3_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_k3_stats -o x -- python scripts/quick_k3tb.py
run ${R}_k3_pmc --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace -d $ROOT/gpurun_out/prof_${R}_k3_pmc -o x -- python scripts/quick_k3tb.py 262144
run ${R}_k2_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_k2_stats -o x -- python scripts/quick_k2c.py
run ${R}_k2_fetch --pmc FETCH_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_k2_fetch -o x -- python scripts/quick_k2c.py
run ${R}_k2_write --pmc WRITE_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_k2_write -o x -- python scripts/quick_k2c.py
run ${R}_k2_pmc_sq --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $ROOT/gpurun_out/prof_${R}_k2_pmc_sq -o x -- python scripts/quick_k2d.py
for t in bench_stats bench_fetch bench_write k1_issue k3_stats k3_pmc k2_stats k2_fetch k2_write k2_pmc_sq; do echo "== $t"; grep -E "polyhip" $ROOT/gpurun_out/${R}_$t.md | head -8 | cut -c1-180; done
