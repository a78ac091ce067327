#!/bin/bash
# round 5, call 5: level 1 with contiguous runs of work items per workgroup, check4 with one part computation per element,
# fine4 at 512 threads (two workgroups per CU) against 1024
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_index_build_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/c05_tests_b4.log
tail -3 gpurun_out/c05_tests_b4.log
timeout 300 python scripts/quick_k2_b4.py sweep > gpurun_out/c05_k2_b4.log 2>&1
cat gpurun_out/c05_k2_b4.log
echo "== fine 512"; POLYHIP_K2_B4_FINE=5 timeout 300 python scripts/quick_k2_b4.py 2>&1 | grep "b4 default"
for sl in 128 64 32; do POLYHIP_K2_B4_SLOTS=$sl bash scripts/collect_profiles_r05.sh k2stats > /dev/null 2>&1; mv gpurun_out/r05_k2_stats.md gpurun_out/c05_k2_stats_slots$sl.md; grep -E "polyhip::k2" gpurun_out/c05_k2_stats_slots$sl.md | head -8 | cut -c1-120; done
POLYHIP_K2_B4_FINE=5 bash scripts/collect_profiles_r05.sh k2stats > /dev/null 2>&1; mv gpurun_out/r05_k2_stats.md gpurun_out/c05_k2_stats_fine512.md; grep -E "polyhip::k2" gpurun_out/c05_k2_stats_fine512.md | head -6 | cut -c1-120
for sl in 64 32; do POLYHIP_K2_B4_SLOTS=$sl bash scripts/collect_profiles_r05.sh k2traffic > /dev/null 2>&1
echo "== traffic slots $sl"; grep -E "scatter4|fine4|check4" gpurun_out/r05_k2_fetch.md gpurun_out/r05_k2_write.md | grep "SIZE" | cut -c1-170; done
