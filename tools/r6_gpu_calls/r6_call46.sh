#!/bin/bash
# round 6, call 46: the full-width parity gates at the TIMED size (16 frames) on the LAST tree: full fine-tuning (every parameter, after the rework of
# its gradient kernels) and the LoRA train-mode gate, device against fp32 CPU autograd
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c46
mkdir -p $O
cd $R
export TMPDIR=/tmp
( T2V_TEST_TRAIN_PARITY_FRAMES=16 timeout 1500 python -m pytest tests/test_gpu_train_parity.py -q -m gpu -s -k "full_fine_tuning_at_full_width" 2>&1 | grep -v "^$" | tail -8 ) > $O/full_finetune_frames_16.txt 2>&1
cat $O/full_finetune_frames_16.txt | cut -c1-600
grep -n "T2V_TEST_TRAIN_PARITY_FRAMES" tests/test_gpu_train_parity.py | head -3
