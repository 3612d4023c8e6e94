#!/bin/bash
set -u
mkdir -p gpurun_out/c19
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -- python $R/bench.py --steps 1 --warmup 1 --graph 0 --clip 0 --cpu-baseline 0 --distill 0 > $R/gpurun_out/c19/prof_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -- python $R/bench.py --steps 1 --warmup 1 --graph 0 --clip 0 --cpu-baseline 0 --distill 0 > $R/gpurun_out/c19/prof_write.log 2>&1
F=$(find /tmp/prof_f -name "*counter_collection.csv" | head -1); W=$(find /tmp/prof_w -name "*counter_collection.csv" | head -1)
echo $F $W; head -3 $F | cut -c1-600; wc -l $F $W
grep -c "gemm_kernel" $F
python $R/tools/pmc_traffic.py $F $W $R/gpurun_out/c19/gemm_traffic.json | head -30
