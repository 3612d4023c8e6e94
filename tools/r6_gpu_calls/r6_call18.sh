#!/bin/bash
# round 6, call 18: the full fine-tuning student at FULL width (tools/full_finetune_time.py) after the row-blocked weight-gradient products
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c18
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python tools/full_finetune_time.py --frames 16 --steps 3 > $O/full_finetune.json 2> $O/full_finetune.err; tail -1 $O/full_finetune.json; tail -4 $O/full_finetune.err
