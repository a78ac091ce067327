#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_distance_gpu.py tests/test_comm_gpu.py -x -q -m gpu --durations=8 2>&1 | tail -14
