#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r03_suite.log
grep -E "passed|failed" gpurun_out/r03_suite.log
python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench.err; echo "bench rc=$?"
bash scripts/collect_profiles_r03.sh
