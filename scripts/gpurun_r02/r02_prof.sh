#!/bin/bash
# rocprofv3 evidence on the final kernels (summaries -> gpurun_out/r02_*.md, copied into profiles/ afterwards)
cd $GRAFT_REPO_ROOT
bash scripts/collect_profiles_r02.sh 2>&1 | tail -90
