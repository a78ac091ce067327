#!/bin/bash
# round 5, call 44: bench.py with the 500 bp SW leg (and its spot check against the oracle)
mkdir -p gpurun_out
( timeout 600 python bench.py ) > gpurun_out/r05_bench_line.json 2> gpurun_out/c44_bench.err; tail -2 gpurun_out/c44_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_line.json").read().strip().splitlines()[-1])
for name in ("smith_waterman_250bp", "smith_waterman_500bp", "smith_waterman_1kb"):
    k = d["extra"][name]
    print(name, {x: (round(k[x], 2) if isinstance(k[x], float) and k[x] < 1e6 else k[x]) for x in ("score_pass_ms", "traceback_ms", "align_one_call_ms", "cell_updates_per_s", "cell_updates_per_s_with_traceback", "cell_updates_per_s_align_one_call", "score_path", "traceback_path")})
print(d["value"], d["ms_per_step"], json.dumps(d["summary"])[:900])
PY
