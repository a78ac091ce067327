#!/bin/bash
# kernel times of configs[3] in one call (score + strings)
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$ROOT/gpurun_out/prof_onecall; rm -rf $out; mkdir -p $out
( cd $ROOT && POLYHIP_SW_OVERLAP=0 rocprofv3 --kernel-trace --stats -d $out -o x -- python scripts/quick_sw_onecall4.py ) > $out/run.log 2>&1
f=$(find $out -name "*results.db" | head -1)
( cd $ROOT && python scripts/rocpd_summary.py $f r06_onecall_stats > gpurun_out/r06_onecall_stats.md; grep polyhip gpurun_out/r06_onecall_stats.md | head -12 | cut -c1-130 )
rm -rf $out
