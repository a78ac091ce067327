#!/bin/bash
# round 3, first call: the new comm / index-part tests, K2 tests, smoke, the 2-rank one-GPU bench, K3 no-overlap profile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_comm_gpu.py tests/test_distance_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r03_first_tests.log
python __graft_entry__.py --smoke > gpurun_out/r03_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r03_smoke.log
BENCH_ONE_GPU_TEST=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 2 --reads 200000 > gpurun_out/r03_bench_2rank.json 2> gpurun_out/r03_bench_2rank.err; echo "rc=$?" >> gpurun_out/r03_bench_2rank.err
tail -5 gpurun_out/r03_first_tests.log; tail -3 gpurun_out/r03_smoke.log; tail -3 gpurun_out/r03_bench_2rank.err; head -c 3000 gpurun_out/r03_bench_2rank.json
