#!/bin/bash
# round 5, call 39: tests/test_traceback_gpu.py on the final build (call 38 stopped at a wrong expectation of a new test in
# this file, after every other file had passed)
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_traceback_gpu.py -m gpu -x -q ) > gpurun_out/c39_traceback_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/c39_traceback_tests.log | tail -3; grep real gpurun_out/c39_traceback_tests.log
