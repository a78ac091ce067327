#!/bin/bash
# counters of the half-float traceback kernel (LDS side)
R=r02
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift
  out=$ROOT/gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
  ( cd $ROOT && rocprofv3 "$@" ) > $out/run.log 2>&1
  f=$(find $out -name "*results.db" | head -1)
  if [ -n "$f" ]; then ( cd $ROOT && python scripts/rocpd_summary.py $f "$tag" > gpurun_out/$tag.md 2>&1 ); else echo "no db for $tag"; tail -5 $out/run.log; fi
  rm -rf $out
}
run ${R}_tbh_pmc_a --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $ROOT/gpurun_out/prof_${R}_tbh_pmc_a -o x -- python scripts/quick_k3tb.py 262144
run ${R}_tbh_pmc_b --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS --kernel-trace -d $ROOT/gpurun_out/prof_${R}_tbh_pmc_b -o x -- python scripts/quick_k3tb.py 262144
grep -h "tb_prof16" $ROOT/gpurun_out/${R}_tbh_pmc_a.md $ROOT/gpurun_out/${R}_tbh_pmc_b.md | cut -c1-160
