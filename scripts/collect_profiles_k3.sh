#!/bin/bash
# The K3-side subset of collect_profiles.sh (kernel stats only): the full default bench command, config 4, and
# the 250-bp / 1-kb read shapes.  Summaries land in gpurun_out/<round>_*.md; copy them into profiles/.
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
run ${R}_bench_full_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_bench_full_stats -o x -- python bench.py --no-cpu-baseline
run ${R}_k3_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_k3_stats -o x -- python scripts/quick_k3tb.py
run ${R}_k3_250bp_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_k3_250bp_stats -o x -- python scripts/quick_k3tb.py 400000 250 5000
run ${R}_k3_1kb_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_k3_1kb_stats -o x -- python scripts/quick_k3tb.py 20000 1000 5000
for t in bench_full_stats k3_stats k3_250bp_stats k3_1kb_stats; do echo "== $t"; grep -E "polyhip" $ROOT/gpurun_out/${R}_$t.md | head -8; done
rm -rf $ROOT/gpurun_out/prof_${R}_*
