#!/bin/bash
# round 5, call 30: is the K1 slab kernel bound by its instruction count?  One chain block of five / fmix32 cut (timing only;
# the truncated hashes change the survivor count, so the checksum and the slab kernel's share are printed too)
mkdir -p gpurun_out
( for a in 0 16 17 0 16 17; do timeout 60 scripts/ubench/k1_ablate_$a; done ) 2>&1 | tee gpurun_out/c30_k1_chain.log
