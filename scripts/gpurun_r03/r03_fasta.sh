#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fasta_gpu.py tests/test_fastq_gpu.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r03_fasta_tests.log
grep -E "passed|failed|Error|assert" gpurun_out/r03_fasta_tests.log | head
python scripts/quick_feeders.py 2>&1 | tail -4

