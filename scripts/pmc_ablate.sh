#!/bin/bash
# VALU / SALU / LDS instruction counts of the K1 ablation binaries (per dispatch of 100k reads)
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for a in 0 2 3 5; do
  out=/tmp/prof_abl$a; rm -rf $out; mkdir -p $out
  ( cd $ROOT && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d $out -o x -- scripts/ubench/k1_ablate_$a ) > $out/run.log 2>&1
  f=$(find $out -name "*results.db" | head -1)
  echo "== PH_ABL=$a"; ( cd $ROOT && python scripts/rocpd_summary.py $f abl | grep -E "sketch_fast" | cut -c1-160 )
done
