#!/bin/bash
# round 4, call 6: the whole GPU suite on the new library (halo conv in the engines, train-mode teacher, C-side replay)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c6
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $O/pytest_gpu.txt
