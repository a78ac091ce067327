#!/bin/bash
# round 5, call 37: validation of the build with the long-read SW paths -- the whole -m gpu suite, smoke(), bench.py, then the
# kernel statistics of the whole bench and the seqhash legs' statistics and counters on this build
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -8 > gpurun_out/c37_gputests.log; cat gpurun_out/c37_gputests.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
( timeout 600 python bench.py ) > gpurun_out/r05_bench_line.json 2> gpurun_out/c37_bench.err; tail -2 gpurun_out/c37_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_line.json").read().strip().splitlines()[-1])
md = d["extra"]["mash_distance"]
print("K1", d["value"], d["ms_per_step"], "K2", md["counts_ms"], md["index_build_ms"], md["join_only_ms"], md["roofline"]["frac"], md["full_matrix_one_gpu"]["ms"], "seqhash", d["extra"]["seqhash"]["ms"])
k = d["extra"]["smith_waterman_1kb"]
print("SW 1kb", {x: k[x] for x in ("score_pass_ms", "traceback_ms", "align_one_call_ms", "cell_updates_per_s", "cell_updates_per_s_with_traceback", "cell_updates_per_s_align_one_call", "score_path", "traceback_path")})
print(json.dumps(d["summary"])[:900])
PY
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
R=r05; ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; out=$ROOT/gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
  ( cd $ROOT && timeout 600 rocprofv3 "$@" ) > $out/run.log 2>&1
  f=$(find $out -name "*results.db" | head -1)
  if [ -n "$f" ]; then ( cd $ROOT && python scripts/rocpd_summary.py $f "$tag" > gpurun_out/$tag.md 2>&1 ); else echo "no db for $tag"; tail -5 $out/run.log; fi
  rm -rf $out; }
run ${R}_bench_full_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_bench_full_stats -o x -- python bench.py --no-cpu-baseline
run ${R}_legsB_stats --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_legsB_stats -o x -- python scripts/quick_legs.py B
run ${R}_legsB_fetch --pmc FETCH_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_legsB_fetch -o x -- python scripts/quick_legs.py B
run ${R}_legsB_write --pmc WRITE_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_${R}_legsB_write -o x -- python scripts/quick_legs.py B
cd $ROOT; grep -E "tb_wave|sw_wave8|sw_pkb|k5|seqhash|s2::" gpurun_out/${R}_bench_full_stats.md | head -12 | cut -c1-150
