#!/bin/bash
# round 5, call 38: the long-read SW paths against the oracle at bench size (20,000 pairs per leg), then the whole -m gpu suite
# on the final build with its full summary kept
mkdir -p gpurun_out
timeout 600 python scripts/sweep_long.py 20000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_sweep_long.log
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/c38_gputests_full.log 2>&1
grep -E "passed|failed|error" gpurun_out/c38_gputests_full.log | tail -3; grep real gpurun_out/c38_gputests_full.log
