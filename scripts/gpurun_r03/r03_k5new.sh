#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_seqhash_gpu.py tests/test_abi_gpu.py -q -m gpu 2>&1 | tail -5
timeout 300 python scripts/fuzz_k5.py 60 1 2>&1 | tail -3
python scripts/quick_k5.py 2>&1 | tail -8
