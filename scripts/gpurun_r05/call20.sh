#!/bin/bash
# round 5, call 20: K1 bottom-s loops unrolled by 4 / 5 / 6 (same sketches word for word: the checksums)
mkdir -p gpurun_out
( for u in 4 5 6 4 5 6; do echo -n "PH_BS_U=$u: "; timeout 120 scripts/ubench/k1_ablate_u$u; done ) > gpurun_out/c20_k1_bsu.log 2>&1
cat gpurun_out/c20_k1_bsu.log
