#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 120 python scripts/quick_k5c.py 2>&1 | grep "L="
timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -3
