#!/bin/bash
# round 5, call 14: seqhash with the reverse complement read where it is needed (no second strand written, one K5 pass for
# both strands): parity tests, fuzzers, timing, kernel stats
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_seqhash_gpu.py tests/test_clone_gpu.py tests/test_stress_gpu.py tests/test_primers_gpu.py -x -q 2>&1 | tail -6 > gpurun_out/c14_tests.log
cat gpurun_out/c14_tests.log
timeout 600 python scripts/fuzz_k5.py > gpurun_out/c14_fuzz_k5.log 2>&1; tail -4 gpurun_out/c14_fuzz_k5.log
timeout 600 python scripts/fuzz_misc.py > gpurun_out/c14_fuzz_misc.log 2>&1; tail -4 gpurun_out/c14_fuzz_misc.log
timeout 200 python scripts/quick_seqhash.py 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_sh; rm -rf $out; mkdir -p $out
( cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats -d $out -o x -- python scripts/quick_seqhash.py ) > $out/run.log 2>&1
f=$(find $out -name "*results.db" | head -1)
( cd $GRAFT_REPO_ROOT && python scripts/rocpd_summary.py $f "r05 seqhash stats" > gpurun_out/r05_seqhash_stats.md 2>&1 ); rm -rf $out
grep -E "polyhip" $GRAFT_REPO_ROOT/gpurun_out/r05_seqhash_stats.md | head -8 | cut -c1-130
