#!/bin/bash
# round 6, call 14: full fine-tuning on the device (row a20): kernels of csrc/full_grad.hip and the engine against the reference's gradient fixture
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c14
mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "im2col or affine_grad" 2>&1 | tail -15 ) > $O/t_kernels.txt 2>&1
tail -5 $O/t_kernels.txt
( timeout 1500 python -m pytest tests/test_gpu_train_parity.py -q -m gpu -s -k "full_fine_tuning" 2>&1 | tail -40 ) > $O/t_full.txt 2>&1
grep -v "^$" $O/t_full.txt | tail -25
