#!/bin/bash
# K1 issue / LDS utilisation counters (two PMC passes, kernel-trace only)
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); out=/tmp/prof_k1b$i; rm -rf $out; mkdir -p $out
  ( cd $ROOT && rocprofv3 --pmc $set --kernel-trace -d $out -o x -- python scripts/quick_k1.py ) > $out/run.log 2>&1
  f=$(find $out -name "*results.db" | head -1)
  if [ -n "$f" ]; then ( cd $ROOT && python scripts/rocpd_summary.py $f k1b$i | grep -E "sketch_fast" | cut -c1-150 ); else tail -5 $out/run.log; fi
done
