#!/bin/bash
# round 5, call 27: index build tests (incl. the repeated-hash list overflow) under a short timeout
timeout 300 python -m pytest tests/test_index_build_gpu.py -x -q 2>&1 | tail -4
