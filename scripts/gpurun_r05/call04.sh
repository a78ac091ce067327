#!/bin/bash
# round 5, call 4: fine4 with leader-aggregated ranks, scan4 vectorised, XCD-aware level 1, the 512 x 32 stage
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_index_build_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/c04_tests_b4.log
timeout 300 python scripts/quick_k2_b4.py sweep > gpurun_out/c04_k2_b4.log 2>&1
tail -3 gpurun_out/c04_tests_b4.log; cat gpurun_out/c04_k2_b4.log
for sl in 128 64 32; do POLYHIP_K2_B4_SLOTS=$sl bash scripts/collect_profiles_r05.sh k2stats > /dev/null 2>&1; mv gpurun_out/r05_k2_stats.md gpurun_out/c04_k2_stats_slots$sl.md; grep -E "polyhip::k2" gpurun_out/c04_k2_stats_slots$sl.md | head -12 | cut -c1-120; done
POLYHIP_K2_B4_SLOTS=32 bash scripts/collect_profiles_r05.sh k2traffic > /dev/null 2>&1
grep -E "scatter4|fine4|check4|rowjoin_dense_kernel<10, true" gpurun_out/r05_k2_fetch.md gpurun_out/r05_k2_write.md | grep "SIZE" | cut -c1-170
