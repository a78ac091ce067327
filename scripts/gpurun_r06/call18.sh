#!/bin/bash
# the packed score pass's sub-batches: equal shares of the batch (default now) against 262,144 pairs (one full round) and others
mkdir -p gpurun_out
{
for sub in "" 262144 250112 200064 131072 125056; do
  echo -n "POLYHIP_SW_SUB=${sub:-default}: "; POLYHIP_SW_SUB=$sub python scripts/quick_k3tb.py 2>&1 | grep "K3 score" | cut -c1-80
done
echo -n "POLYHIP_SW_OVERLAP=0: "; POLYHIP_SW_OVERLAP=0 python scripts/quick_k3tb.py 2>&1 | grep "K3 score" | cut -c1-80
} > gpurun_out/r06_k3_subbatch.log 2>&1
cat gpurun_out/r06_k3_subbatch.log
