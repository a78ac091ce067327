#!/bin/bash
# round 5, call 28: bench.py --gpus 2 with both ranks on GPU 0 (gloo): the N-rank legs after the round's changes
mkdir -p gpurun_out
( time BENCH_ONE_GPU_TEST=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --reads 200000 ) > gpurun_out/r05_bench_2rank_onegpu_test.json 2> gpurun_out/c28.err
tail -5 gpurun_out/c28.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_2rank_onegpu_test.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d["value"], d["launch"])
print(json.dumps(d.get("extra", {}).get("mash_distance_allgather"), indent=0)[:1500])
PY
