#!/bin/bash
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
tag=r03_k2_pmc_lds
out=$ROOT/gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
( cd $ROOT && timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace -d $out -o x -- python scripts/quick_k2c.py ) > $out/run.log 2>&1
f=$(find $out -name "*results.db" | head -1)
if [ -n "$f" ]; then ( cd $ROOT && timeout 100 python scripts/rocpd_summary.py $f "$tag" > gpurun_out/$tag.md 2>&1 ); else echo "no db"; tail -5 $out/run.log; fi
rm -rf $out
grep "rowjoin_dense_kernel<10, true>" $ROOT/gpurun_out/$tag.md | cut -c1-150
