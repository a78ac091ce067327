#!/bin/bash
cd $GRAFT_REPO_ROOT
{
echo "# randomised sweeps on the round's final build (scripts/fuzz_r03.py, fuzz_k5.py, fuzz_k3.py; seeds differ from the earlier logs)"
timeout 120 python scripts/fuzz_r03.py 75 20260924 2>&1 | tail -2
timeout 100 python scripts/fuzz_k5.py 60 20260924 2>&1 | tail -1
timeout 120 python scripts/fuzz_k3.py 60 2>&1 | tail -2
} > gpurun_out/r03_fuzz_final.log 2>&1
cat gpurun_out/r03_fuzz_final.log
