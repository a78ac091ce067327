#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_traceback_gpu.py -x -q -m gpu -k "packed or chunk_pipeline or TestSmith or examples" 2>&1 | tail -6
python - <<'PY'
import sys, json, torch
sys.path.insert(0, '.')
from poly_amd import bench_extra
r = bench_extra.e2e(torch.device('cuda:0'))
for k in ("smith_waterman", "smith_waterman_with_strings", "smith_waterman_with_packed_strings"):
    print(k, {a: b for a, b in r[k].items() if a != "workload"})
PY
