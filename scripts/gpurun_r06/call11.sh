#!/bin/bash
# randomised sweeps over the round's new SW kernels (sw_pk1x2_kernel, the 64-row tiles) and the rest
mkdir -p gpurun_out
timeout 900 python scripts/fuzz_k3.py > gpurun_out/r06_fuzz_k3.log 2>&1; echo "fuzz_k3 rc=$?"; tail -3 gpurun_out/r06_fuzz_k3.log
timeout 600 python scripts/fuzz_misc.py > gpurun_out/r06_fuzz_misc.log 2>&1; echo "fuzz_misc rc=$?"; tail -3 gpurun_out/r06_fuzz_misc.log
