#!/bin/bash
cd $GRAFT_REPO_ROOT
( timeout 400 python scripts/fuzz_r03.py 150 301 2>&1 | grep -v amdgpu | tail -6 ) > gpurun_out/r03_fuzz.log
( timeout 300 python scripts/fuzz_k1.py 60 911 2>&1 | tail -2 ) >> gpurun_out/r03_fuzz.log
( timeout 400 python scripts/fuzz_misc.py 90 733 2>&1 | tail -2 ) >> gpurun_out/r03_fuzz.log
cat gpurun_out/r03_fuzz.log
