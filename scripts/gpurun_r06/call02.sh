#!/bin/bash
# K1 round 6: the windowed rank pass (PH_BS_WIN) against the whole-bin rank; probes 21 / 22; lever (a) PH_SELBINS; LDS padding
mkdir -p gpurun_out
{
for i in 1 2; do
for v in base win0 win3 win0_nobar win4_nobar win0_norank win4_norank nobs selb11 selb10 pad4k pad8k; do
  echo -n "$v: "; timeout 120 scripts/ubench/k1_v_$v
done
done
} > gpurun_out/r06_k1_variants.log 2>&1
timeout 900 python -m pytest tests/test_mash_gpu.py -x -q 2>&1 | tail -3 > gpurun_out/r06_k1_tests.log
timeout 600 python scripts/fuzz_k1.py > gpurun_out/r06_fuzz_k1.log 2>&1; tail -3 gpurun_out/r06_fuzz_k1.log
cat gpurun_out/r06_k1_variants.log gpurun_out/r06_k1_tests.log
