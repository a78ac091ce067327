#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_align_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
echo "== pk1"; python scripts/quick_k3tb.py 2>&1 | grep "K3 score"
echo "== POLYHIP_SW_PK1=0"; POLYHIP_SW_PK1=0 python scripts/quick_k3tb.py 2>&1 | grep "K3 score"
