#!/bin/bash
# round 5, call 29: final validation -- the whole -m gpu suite, smoke(), then the evidence set on the final build
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -8 > gpurun_out/c29_gputests.log; cat gpurun_out/c29_gputests.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
( timeout 600 python bench.py ) > gpurun_out/r05_bench_line.json 2> gpurun_out/c29_bench.err; tail -2 gpurun_out/c29_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_line.json").read().strip().splitlines()[-1])
md = d["extra"]["mash_distance"]
print("K1", d["value"], d["ms_per_step"], "K2", md["counts_ms"], md["index_build_ms"], md["join_only_ms"], md["roofline"]["frac"], md["full_matrix_one_gpu"]["ms"], "seqhash", d["extra"]["seqhash"]["ms"])
print(json.dumps(d["summary"])[:700])
PY
