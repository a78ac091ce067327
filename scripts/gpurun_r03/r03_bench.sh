#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench.err; echo "rc=$?" >> gpurun_out/r03_bench.err
tail -3 gpurun_out/r03_bench.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r03_bench_line.json'))
print({k:j[k] for k in ('value','ms_per_step')}, j['roofline']['frac'])
for k,v in j.get('extra',{}).items():
    print(k, json.dumps(v)[:600])
PY
