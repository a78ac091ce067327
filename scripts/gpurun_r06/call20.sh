#!/bin/bash
# near ties (every block worth M within three blocks of the first) resolved by sw_locate16_kernel instead of the full sweep
mkdir -p gpurun_out
{
python scripts/quick_k3tb.py 2>&1 | grep "K3 score"
python scripts/quick_k3tb.py 2>&1 | grep "K3 score"
} | tee gpurun_out/r06_k3_near_ties.log
timeout 1500 python -m pytest tests/test_align_gpu.py tests/test_traceback_gpu.py tests/test_nw_gpu.py -x -q 2>&1 | grep -E "passed|failed|error" | tee -a gpurun_out/r06_k3_near_ties.log
timeout 700 python scripts/fuzz_k3.py > gpurun_out/r06_fuzz_k3_near.log 2>&1; echo "fuzz_k3 rc=$?"; tail -2 gpurun_out/r06_fuzz_k3_near.log
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_nt -o x -- python $GRAFT_REPO_ROOT/scripts/quick_k3tb.py > /tmp/prof_nt.log 2>&1; f=$(find /tmp/prof_nt -name "*results.db" | head -1); cd $GRAFT_REPO_ROOT; [ -n "$f" ] && python scripts/rocpd_summary.py $f r06_k3_near_ties_stats | grep polyhip | head -6
