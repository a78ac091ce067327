#!/bin/bash
# round-2 first GPU pass: the whole -m gpu suite, smoke, the C harness, the default bench line, and the
# self-launching 2-rank dry run (both ranks on GPU 0 over gloo)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r02_build.log 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r02_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02_tests.log
tail -5 gpurun_out/r02_tests.log
timeout 120 ./tests/abi/abi_smoke tests/golden/puc19.seq > gpurun_out/r02_abi_smoke.log 2>&1; echo "abi rc=$?" | tee -a gpurun_out/r02_abi_smoke.log
cat gpurun_out/r02_abi_smoke.log
timeout 600 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; echo "bench rc=$?"
head -c 1500 gpurun_out/r02_bench.json; tail -3 gpurun_out/r02_bench.err
BENCH_ONE_GPU_TEST=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --reads 200000 > gpurun_out/r02_bench_2rank_onegpu.json 2> gpurun_out/r02_bench_2rank_onegpu.err; echo "2-rank rc=$?"
head -c 3000 gpurun_out/r02_bench_2rank_onegpu.json; tail -5 gpurun_out/r02_bench_2rank_onegpu.err
python bench.py --gpus 2 --steps 2 --warmup 1 > /dev/null 2> gpurun_out/r02_bench_refuse.err; echo "2 ranks on 1 GPU without the test flag: rc=$? (expected non-zero)"; tail -2 gpurun_out/r02_bench_refuse.err
