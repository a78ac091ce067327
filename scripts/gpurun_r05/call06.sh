#!/bin/bash
# round 5, call 6: fine4 with its loads back to back (buffer loads) and branch-free stores; level 1 back to round robin
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_index_build_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/c06_tests_b4.log
tail -3 gpurun_out/c06_tests_b4.log
timeout 300 python scripts/quick_k2_b4.py > gpurun_out/c06_k2_b4.log 2>&1
cat gpurun_out/c06_k2_b4.log
echo "== fine 512"; POLYHIP_K2_B4_FINE=5 timeout 300 python scripts/quick_k2_b4.py 2>&1 | grep "b4 default"
bash scripts/collect_profiles_r05.sh k2stats > /dev/null 2>&1; mv gpurun_out/r05_k2_stats.md gpurun_out/c06_k2_stats.md; grep -E "polyhip::k2" gpurun_out/c06_k2_stats.md | head -24 | cut -c1-120
POLYHIP_K2_B4_FINE=5 bash scripts/collect_profiles_r05.sh k2stats > /dev/null 2>&1; mv gpurun_out/r05_k2_stats.md gpurun_out/c06_k2_stats_fine512.md; grep -E "polyhip::k2" gpurun_out/c06_k2_stats_fine512.md | head -4 | cut -c1-120
