set -x
mkdir -p gpurun_out
rm -f gpurun_out/k3_long.log
for args in "20000 1000 5000" "60000 300 5000" "10000 2000 5000"; do
  echo "== path 7: $args" >> gpurun_out/k3_long.log
  timeout 300 python scripts/quick_k3tb.py $args 2>&1 | grep K3 >> gpurun_out/k3_long.log
  echo "== POLYHIP_SW_PACKED=0: $args" >> gpurun_out/k3_long.log
  POLYHIP_SW_PACKED=0 timeout 300 python scripts/quick_k3tb.py $args 2>&1 | grep K3 >> gpurun_out/k3_long.log
done
timeout 900 python -m pytest tests/test_align_gpu.py tests/test_traceback_gpu.py -m gpu -x -q 2>&1 | tail -15 >> gpurun_out/k3_long.log
cat gpurun_out/k3_long.log
