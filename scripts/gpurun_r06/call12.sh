#!/bin/bash
# the round's evidence in one command (scripts/collect_profiles_r06.sh), then the bench line and the GPU test log of the same build
mkdir -p gpurun_out
bash scripts/collect_profiles_r06.sh 2>&1 | tail -60 > gpurun_out/r06_collect.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err; echo "bench rc=$? $(wc -c < gpurun_out/r06_bench_line.json) bytes"
cp gpurun_out/bench_extra.json gpurun_out/r06_bench_full.json 2>/dev/null
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r06_gpu_tests_final.log; cat gpurun_out/r06_gpu_tests_final.log
tail -40 gpurun_out/r06_collect.log
