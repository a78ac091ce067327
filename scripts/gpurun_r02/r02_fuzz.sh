#!/bin/bash
# randomised sweeps with fresh seeds after the half-float cell / K2 walk + index rework
cd $GRAFT_REPO_ROOT
( timeout 200 python scripts/fuzz_k3.py 100 777000 2>&1 | tail -2
  timeout 200 python scripts/fuzz_misc.py 100 888000 2>&1 | tail -2
  timeout 200 python scripts/fuzz_k1.py 70 999000 2>&1 | tail -2 ) | grep -v amdgpu.ids | tee gpurun_out/r02_fuzz.log
