#!/bin/bash
# randomised sweeps after the half-float traceback / locate and the two-stream overlaps
cd $GRAFT_REPO_ROOT
( timeout 260 python scripts/fuzz_k3.py 170 31337 2>&1 | tail -2
  timeout 120 python scripts/fuzz_misc.py 60 424242 2>&1 | tail -2 ) | grep -v amdgpu.ids | tee gpurun_out/r02_fuzz2.log
