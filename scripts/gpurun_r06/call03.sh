#!/bin/bash
# round 6: sw_pk1x2_kernel (two lanes per lane's rows) against sw_pk1_kernel, + second batch of K1 variants
mkdir -p gpurun_out
{
echo "== K3 default (pk1x2)"; python scripts/quick_k3tb.py 2>&1 | grep "K3 score"
echo "== K3 POLYHIP_SW_PK1X2=0"; POLYHIP_SW_PK1X2=0 python scripts/quick_k3tb.py 2>&1 | grep "K3 score"
echo "== K3 default (pk1x2) again"; python scripts/quick_k3tb.py 2>&1 | grep "K3 score"
} > gpurun_out/r06_k3_x2.log 2>&1
cat gpurun_out/r06_k3_x2.log
timeout 1200 python -m pytest tests/test_align_gpu.py -x -q -k "packed_pass_equals or config4 or half_float" 2>&1 | tail -5 | tee gpurun_out/r06_k3_x2_tests.log
{
for i in 1 2; do
for v in base win5 win6 win8 sig5 sig4 sig3 selat selat_sig4 bsu5 selat_nobs nobs; do
  echo -n "$v: "; timeout 120 scripts/ubench/k1_v_$v
done
done
} > gpurun_out/r06_k1_variants2.log 2>&1
cat gpurun_out/r06_k1_variants2.log
