#!/bin/bash
# round 6, call 45: t2v_wgrad_tn_conv (a conv leaf's weight gradient stored in the parameter's layout: no tap-major temporary, no gather): tests, step time, parity tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c45
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_unet_grad.py -q -x -m gpu -k "wgrad" 2>&1 | tail -3 | tee $O/pytest_wgrad.txt
RUN=r6c45 bash tools/r6_gpu_calls/r6_call37.sh
