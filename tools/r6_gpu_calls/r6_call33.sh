#!/bin/bash
# round 6, call 33: where the full fine-tuning step sits after the final-kernel fix: rocprofv3 kernel summary of tools/full_finetune_time.py (4 steps after the recording pass)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c33
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_ff
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ff -- python $R/tools/full_finetune_time.py --frames 16 --steps 4 > $O/ff.log 2>&1
cp $(find /tmp/prof_ff -name "*kernel_stats.csv" | head -1) $O/full_finetune_kernel_stats.csv
grep "^{" $O/ff.log | tail -1 | cut -c1-400
head -25 $O/full_finetune_kernel_stats.csv | cut -c1-160
