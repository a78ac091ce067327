#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 | tee gpurun_out/r02_tests_all.log
timeout 600 python bench.py > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.err; echo "bench rc=$?"
bash scripts/collect_profiles_r02.sh 2>&1 | tail -70
