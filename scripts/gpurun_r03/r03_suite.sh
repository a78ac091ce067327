#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r03_suite.log
cat gpurun_out/r03_suite.log
