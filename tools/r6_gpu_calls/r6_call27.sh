#!/bin/bash
# round 6, call 27: the full fine-tuning step at full width, native gradient engine vs the torch composite path (ATen kernels) on the same box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c27
mkdir -p $O
cd $R
export TMPDIR=/tmp
for n in 1 0; do
  timeout 1200 python tools/full_finetune_time.py --frames 16 --steps 3 --native $n > $O/full_finetune_native$n.json 2> $O/err_$n.txt; tail -1 $O/full_finetune_native$n.json | cut -c1-600; grep -i "error\|Traceback" -A3 $O/err_$n.txt | head -8
done
