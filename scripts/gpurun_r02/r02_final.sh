#!/bin/bash
# full -m gpu suite, smoke, default bench, and the 2-rank launch test sharing GPU 0 (gloo)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 | tee gpurun_out/r02_tests_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/r02_smoke_final.log
timeout 600 python bench.py > gpurun_out/r02_final_bench.json 2> gpurun_out/r02_final_bench.err; echo "bench rc=$?"
BENCH_ONE_GPU_TEST=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r02_final_bench2.json 2> gpurun_out/r02_final_bench2.err; echo "bench2 rc=$?"
tail -c 600 gpurun_out/r02_final_bench.json
