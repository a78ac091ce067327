#!/bin/bash
# L2 hit rate of the reworked dense join
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$ROOT/gpurun_out/prof_r02_k2_pmc_tcc; rm -rf $out; mkdir -p $out
( cd $ROOT && rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $out -o x -- python scripts/quick_k2d.py ) > $out/run.log 2>&1
f=$(find $out -name "*results.db" | head -1)
( cd $ROOT && python scripts/rocpd_summary.py $f r02_k2_pmc_tcc > gpurun_out/r02_k2_pmc_tcc.md 2>&1 )
rm -rf $out
grep -h "rowjoin_dense" $ROOT/gpurun_out/r02_k2_pmc_tcc.md | cut -c1-140
