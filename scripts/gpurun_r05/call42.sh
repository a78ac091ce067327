#!/bin/bash
# round 5, call 42: the build with the int16 block maximum: tests/test_traceback_gpu.py (every test that runs a packed score pass
# in front of a traceback), smoke(), bench.py
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_traceback_gpu.py -m gpu -x -q ) > gpurun_out/c42_traceback_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/c42_traceback_tests.log | tail -3
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
( timeout 600 python bench.py ) > gpurun_out/r05_bench_line.json 2> gpurun_out/c42_bench.err; tail -2 gpurun_out/c42_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_line.json").read().strip().splitlines()[-1])
md = d["extra"]["mash_distance"]
print("K1", d["value"], d["ms_per_step"], "K2", md["counts_ms"], md["index_build_ms"], md["join_only_ms"], md["roofline"]["frac"], md["full_matrix_one_gpu"]["ms"], "seqhash", d["extra"]["seqhash"]["ms"])
k = d["extra"]["smith_waterman_1kb"]
print("SW 1kb", {x: k[x] for x in ("score_pass_ms", "traceback_ms", "align_one_call_ms", "cell_updates_per_s", "cell_updates_per_s_with_traceback", "cell_updates_per_s_align_one_call", "score_path", "traceback_path")})
print(json.dumps(d["summary"])[:900])
PY
