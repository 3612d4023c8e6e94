#!/bin/bash
set -u
mkdir -p gpurun_out/c32
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c32 -- python $R/tools/distill_bench.py --steps 2 --warmup 2 --native-student 1 > $R/gpurun_out/c32/prof.log 2>&1
python $R/tools/trace_window.py /tmp/prof_c32 --marker sinh --steps 2 --out $R/gpurun_out/c32/distill_window_stats.csv
head -30 $R/gpurun_out/c32/distill_window_stats.csv | cut -c1-180
