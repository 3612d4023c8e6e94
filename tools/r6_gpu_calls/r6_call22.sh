#!/bin/bash
# round 6, call 22: full fine-tuning at the mid width (model_channels 128) on the device against the reference's gradient fixture
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c22
mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_train_parity.py -q -m gpu -s -k "full_fine_tuning" 2>&1 | grep -v "^$" | tail -12 ) > $O/t_full.txt 2>&1
cat $O/t_full.txt | cut -c1-300
