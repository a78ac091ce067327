#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fastq_gpu.py tests/test_fasta_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -8 | tee gpurun_out/r02_feed_tests.log
python scripts/quick_feeders.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_feed.log
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_feed; rm -rf $out; mkdir -p $out
( cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats -d $out -o x -- python scripts/quick_feeders.py ) > $out/run.log 2>&1
f=$(find $out -name "*results.db" | head -1)
( cd $GRAFT_REPO_ROOT && python scripts/rocpd_summary.py $f r02_feeders > gpurun_out/r02_feeders_stats.md 2>&1 ); grep "fq::" $GRAFT_REPO_ROOT/gpurun_out/r02_feeders_stats.md | head -20
rm -rf $out
