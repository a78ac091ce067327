#!/bin/bash
# FINAL build of round 6: the whole evidence set in one go -- collect_profiles_r06.sh (all sets), smoke, GPU tests, the bench
# line (+ full record), the 2-rank shape on one GPU, the alignment fuzz over the final kernels
mkdir -p gpurun_out
bash scripts/collect_profiles_r06.sh 2>&1 | tail -80 > gpurun_out/r06_collect.log
python __graft_entry__.py --smoke 2>&1 | grep -E "poly_amd.build|smoke ok" | tee gpurun_out/r06_smoke.log
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tee gpurun_out/r06_gpu_tests_final.log
cp gpurun_out/traffic.json gpurun_out/k1_issue.json gpurun_out/k3_issue.json gpurun_out/k1_traffic.json profiles/ 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err; echo "bench rc=$? $(wc -c < gpurun_out/r06_bench_line.json) bytes"
cp gpurun_out/bench_extra.json gpurun_out/r06_bench_full.json 2>/dev/null
BENCH_ONE_GPU_TEST=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --reads 200000 > gpurun_out/r06_bench_2rank_onegpu_test.json 2> gpurun_out/r06_bench_2rank.err; echo "2-rank rc=$?"
timeout 700 python scripts/fuzz_k3.py > gpurun_out/r06_fuzz_k3.log 2>&1; echo "fuzz_k3 rc=$?"; tail -2 gpurun_out/r06_fuzz_k3.log
python scripts/quick_sw_tile64.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_sw_tile64_near_ties.log; cat gpurun_out/r06_sw_tile64_near_ties.log | cut -c1-160
python scripts/quick_k3tb.py 2>&1 | grep "K3 score" > gpurun_out/r06_k3_final.log; POLYHIP_SW_PK1X2=0 python scripts/quick_k3tb.py 2>&1 | grep "K3 score" | sed 's/^/POLYHIP_SW_PK1X2=0: /' >> gpurun_out/r06_k3_final.log; cat gpurun_out/r06_k3_final.log | cut -c1-120
grep -E "sw_pk1x2|sketch_slab" gpurun_out/r06_collect.log | head
