#!/bin/bash
# round 6, call 26: full fine-tuning at FULL width on the device against fp32 CPU autograd, every parameter (2 frames in the suite; then 16 as evidence)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c26
mkdir -p $O
cd $R
export TMPDIR=/tmp
for f in 2 16; do
  ( T2V_TEST_TRAIN_PARITY_FRAMES=$f timeout 1500 python -m pytest tests/test_gpu_train_parity.py -q -m gpu -s -k "full_fine_tuning_at_full_width" 2>&1 | grep -v "^$" | tail -8 ) > $O/frames_$f.txt 2>&1
  cat $O/frames_$f.txt | cut -c1-400
done
