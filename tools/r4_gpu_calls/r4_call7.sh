#!/bin/bash
# round 4, call 7: the failing full-width training parity gate with the training forward back on t2v_gemm; rest of the suite after it
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c7
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_train_parity.py -q -x -s 2>&1 | grep -E "full width|passed|failed|Error|assert" | tail -12 | tee $O/train_parity.txt
timeout 1500 python -m pytest tests/test_gpu_unet_grad.py tests/test_gpu_gemm_fuse.py -q -x 2>&1 | tail -3 | tee -a $O/train_parity.txt
