#!/bin/bash
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift
  out=$ROOT/gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
  ( cd $ROOT && rocprofv3 "$@" ) > $out/run.log 2>&1
  f=$(find $out -name "*results.db" | head -1)
  if [ -n "$f" ]; then ( cd $ROOT && python scripts/rocpd_summary.py $f "$tag" > gpurun_out/$tag.md 2>&1 ); else echo "no db for $tag"; tail -5 $out/run.log; fi
  rm -rf $out; }
run r03_k2idx_pmc_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --kernel-trace -d $ROOT/gpurun_out/prof_r03_k2idx_pmc_lds -o x -- python scripts/quick_k2_index.py
run r03_k2idx_pmc_ta --pmc TA_BUSY_avr TA_FLAT_WRITE_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum --kernel-trace -d $ROOT/gpurun_out/prof_r03_k2idx_pmc_ta -o x -- python scripts/quick_k2_index.py
grep "fine_kernel\|coarse_scatter_staged" $ROOT/gpurun_out/r03_k2idx_pmc_lds.md $ROOT/gpurun_out/r03_k2idx_pmc_ta.md | cut -d: -f2 | cut -c1-150
