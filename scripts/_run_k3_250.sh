set -x
mkdir -p gpurun_out
rm -f gpurun_out/k3_250.log
for args in "400000 250 5000" "400000 200 5000" "1000000 150 5000"; do
  echo "== packed: $args" >> gpurun_out/k3_250.log
  timeout 300 python scripts/quick_k3tb.py $args 2>&1 | grep K3 >> gpurun_out/k3_250.log
done
echo "== POLYHIP_SW_PACKED=0: 400000 250 5000" >> gpurun_out/k3_250.log
POLYHIP_SW_PACKED=0 timeout 300 python scripts/quick_k3tb.py 400000 250 5000 2>&1 | grep K3 >> gpurun_out/k3_250.log
timeout 900 python -m pytest tests/test_align_gpu.py tests/test_traceback_gpu.py -m gpu -x -q 2>&1 | tail -5 >> gpurun_out/k3_250.log
cat gpurun_out/k3_250.log
