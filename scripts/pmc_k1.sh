#!/bin/bash
# SQ counter passes over scripts/quick_k1.py (K1 at 100k reads); summaries -> gpurun_out/<tag>_pmc_k1_*.md
R=${1:-r01}
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
pass() { tag=$1; shift
  out=/tmp/prof_$tag; rm -rf $out; mkdir -p $out
  ( cd $ROOT && rocprofv3 --pmc "$@" --kernel-trace -d $out -o x -- python scripts/quick_k1.py ) > $out/run.log 2>&1
  f=$(find $out -name "*results.db" | head -1)
  if [ -n "$f" ]; then ( cd $ROOT && python scripts/rocpd_summary.py $f "$tag" | grep -E "sketch_fast|failed" | cut -c1-160 > gpurun_out/$tag.md ); cat $ROOT/gpurun_out/$tag.md; else echo "no db for $tag"; grep -iE "error|invalid|not" $out/run.log | head -5; fi
}
pass ${R}_pmc_k1_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass ${R}_pmc_k1_b SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM


