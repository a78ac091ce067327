#!/bin/bash
# full GPU suite, the bench line, the rocprofv3 summaries of round 3 (copy gpurun_out/r03_*.md / .json into profiles/)
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r03_suite.log
grep -E "passed|failed" gpurun_out/r03_suite.log
python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench.err; echo "bench rc=$?"
bash scripts/collect_profiles_r03.sh > gpurun_out/r03_collect.log 2>&1
bash scripts/gpurun_r03/r03_k3_nooverlap.sh > gpurun_out/r03_k3_collect.log 2>&1
tail -12 gpurun_out/r03_k3_collect.log
python scripts/quick_k5.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_k5_quick.log
python scripts/quick_k5b.py 2>&1 | grep "L=" >> gpurun_out/r03_k5_quick.log
timeout 200 python scripts/fuzz_k5.py 60 7 2>&1 | tail -1 >> gpurun_out/r03_k5_quick.log
