#!/bin/bash
# the two-lane score pass on shorter tiles (2 x 32 / 48 / 64 rows): reads of 64 / 96 / 100 / 125 bp against the one-lane kernel
mkdir -p gpurun_out
{
for LA in 64 96 100 125 150; do
  echo -n "LA=$LA two lanes: "; python scripts/quick_k3tb.py 1000000 $LA 5000 2>&1 | grep "K3 score" | cut -c1-60
  echo -n "LA=$LA one lane : "; POLYHIP_SW_PK1X2=0 python scripts/quick_k3tb.py 1000000 $LA 5000 2>&1 | grep "K3 score" | cut -c1-60
done
} > gpurun_out/r06_k3_x2_tiles.log 2>&1
cat gpurun_out/r06_k3_x2_tiles.log
timeout 1500 python -m pytest tests/test_align_gpu.py tests/test_traceback_gpu.py -x -q 2>&1 | grep -E "passed|failed|error" | tee gpurun_out/r06_k3_x2_tiles_tests.log
