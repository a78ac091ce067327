#!/bin/bash
# final build: GPU tests, smoke, the bench line (1 rank), and the 2-rank shape on one GPU (gloo; exercises the N > 1 code)
mkdir -p gpurun_out
python __graft_entry__.py --smoke 2>&1 | grep -E "poly_amd.build|smoke ok" | tee gpurun_out/r06_smoke.log
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tee gpurun_out/r06_gpu_tests_final.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err; echo "bench rc=$? $(wc -c < gpurun_out/r06_bench_line.json) bytes"
cp gpurun_out/bench_extra.json gpurun_out/r06_bench_full.json 2>/dev/null
BENCH_ONE_GPU_TEST=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --reads 200000 > gpurun_out/r06_bench_2rank_onegpu_test.json 2> gpurun_out/r06_bench_2rank.err; echo "2-rank rc=$? $(wc -c < gpurun_out/r06_bench_2rank_onegpu_test.json) bytes"
