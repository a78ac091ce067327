#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun):
#   gpurun_out/<round>_bench_stats.md   kernel-trace stats of the default bench command
#   gpurun_out/<round>_bench_fetch.md / _write.md   PMC passes (separate runs, kernel-trace only)
#   gpurun_out/<round>_calib_fetch.md / _write.md   FETCH_SIZE / WRITE_SIZE calibration on known byte counts
# Copy the summaries into profiles/ afterwards.
R=${1:-r01}
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { # tag, rocprof args..., -- cmd
  tag=$1; shift
  out=$ROOT/gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
  ( cd $ROOT && rocprofv3 "$@" ) > $out/run.log 2>&1
  f=$(find $out -name "*results.db" | head -1)
  if [ -n "$f" ]; then ( cd $ROOT && python scripts/rocpd_summary.py $f "$tag" > gpurun_out/$tag.md 2>&1 ); else echo "no db for $tag"; tail -5 $out/run.log; fi
}
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
# kernel stats of the headline command (default steps/warmup, without the secondary kernels so that the
# K1 average is the timed region's), and of the full default command
run ${R}_bench_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_bench_stats -o x -- python bench.py --no-extra --no-cpu-baseline
run ${R}_bench_full_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_bench_full_stats -o x -- python bench.py --no-cpu-baseline
run ${R}_bench_fetch --pmc FETCH_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_bench_fetch -o x -- $BENCH
run ${R}_bench_write --pmc WRITE_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_bench_write -o x -- $BENCH
run ${R}_calib_fetch --pmc FETCH_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_calib_fetch -o x -- scripts/ubench/hbm_calib
run ${R}_calib_write --pmc WRITE_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_calib_write -o x -- scripts/ubench/hbm_calib
# K3 (co-headline): kernel stats of score pass + traceback at config 4, and the issue counters of the packed pass
run ${R}_k3_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_k3_stats -o x -- python scripts/quick_k3tb.py
run ${R}_k3_pmc --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d $ROOT/gpurun_out/prof_${R}_k3_pmc -o x -- python scripts/quick_k3tb.py 262144
for t in bench_stats bench_full_stats bench_fetch bench_write calib_fetch calib_write k3_stats k3_pmc; do echo "== $t"; grep -E "sketch_fast_kernel|sketch_general|read4|read16|write4|write8|write16|failed|sw_pk|sw_locate|sw_wave|tb_prof" $ROOT/gpurun_out/${R}_$t.md | head -12; done
rm -rf $ROOT/gpurun_out/prof_${R}_*   # keep the summaries, drop the databases
