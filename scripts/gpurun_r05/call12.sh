#!/bin/bash
# round 5, call 12: phase ablation of the K1 SLAB kernel (scripts/ubench/k1_ablate.hip, PH_ABL 0 / 11 / 12 / 14 / 15):
# time per 100k reads and VALU instructions per variant
mkdir -p gpurun_out
( for a in 0 11 12 14 15; do timeout 120 scripts/ubench/k1_ablate_$a; done ) > gpurun_out/r05_k1_slab_ablation.log 2>&1
cat gpurun_out/r05_k1_slab_ablation.log
cd /tmp && export TMPDIR=/tmp
for a in 0 11 12 14 15; do
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_k1abl_$a; rm -rf $out; mkdir -p $out
  ( cd $GRAFT_REPO_ROOT && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $out -o x -- scripts/ubench/k1_ablate_$a ) > $out/run.log 2>&1
  f=$(find $out -name "*results.db" | head -1)
  if [ -n "$f" ]; then ( cd $GRAFT_REPO_ROOT && python scripts/rocpd_summary.py $f "k1 slab ablation PH_ABL=$a" > gpurun_out/r05_k1_slab_abl_pmc_$a.md 2>&1 ); fi
  rm -rf $out
  grep -E "sketch_slab_kernel" $GRAFT_REPO_ROOT/gpurun_out/r05_k1_slab_abl_pmc_$a.md | cut -c1-150
done
