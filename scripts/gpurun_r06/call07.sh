#!/bin/bash
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r06_gpu_tests_mid.log; cat gpurun_out/r06_gpu_tests_mid.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench1.out 2> gpurun_out/r06_bench1.err; echo rc=$?; wc -c gpurun_out/r06_bench1.out
