#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 700 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 300 python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench.err; echo "bench rc=$?"
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
