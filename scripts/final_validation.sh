mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed" > gpurun_out/final_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> gpurun_out/final_tests.log
timeout 900 python bench.py 2> gpurun_out/bench_final.err | tail -1 > gpurun_out/bench_final.json
cat gpurun_out/final_tests.log; head -c 1500 gpurun_out/bench_final.json; echo
timeout 60 python scripts/fuzz_k3.py 25 777 2>&1 | tail -2
timeout 900 bash scripts/collect_profiles_k3.sh r01 2>&1 | tail -40
