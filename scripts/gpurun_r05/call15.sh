#!/bin/bash
# round 5, call 15: the whole -m gpu suite, smoke(), and a driver-style bench line
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -12 > gpurun_out/c15_gputests.log; cat gpurun_out/c15_gputests.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
( time timeout 600 python bench.py ) > gpurun_out/c15_bench.json 2> gpurun_out/c15_bench.err; tail -3 gpurun_out/c15_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c15_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
md = d["extra"]["mash_distance"]
print({k: md[k] for k in ("counts_ms", "index_build_ms", "join_only_ms", "index_build")}, md["roofline"]["frac"], md["roofline"].get("traffic_ratio"), md["full_matrix_one_gpu"]["ms"])
print("seqhash", d["extra"]["seqhash"]["ms"], "summary", json.dumps(d["summary"])[:600])
PY
