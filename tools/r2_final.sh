#!/bin/bash
# Round-2 closing GPU call: the whole -m gpu suite, the default bench line, the rocprofv3 kernel summary of the same step and the
# two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only) behind roofline.traffic.
set -u
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/final/pytest_gpu.txt 2>&1; tail -6 gpurun_out/final/pytest_gpu.txt | cut -c1-300
timeout 900 python bench.py > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/final/bench.err | cut -c1-300
timeout 300 python tools/widen_bench.py > gpurun_out/final/widen_bench.json 2> gpurun_out/final/widen.err; tail -c 600 gpurun_out/final/widen_bench.json
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --steps 10 --warmup 2 --clip 0 --cpu-baseline 0 --distill 0 --graph 0 > $R/gpurun_out/final/prof_stats.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/final/unet_bench_kernel_stats.csv
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -- python $R/bench.py --steps 1 --warmup 1 --graph 0 --clip 0 --cpu-baseline 0 --distill 0 > $R/gpurun_out/final/prof_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -- python $R/bench.py --steps 1 --warmup 1 --graph 0 --clip 0 --cpu-baseline 0 --distill 0 > $R/gpurun_out/final/prof_write.log 2>&1
python $R/tools/pmc_traffic.py $(find /tmp/prof_f -name "*counter_collection.csv" | head -1) $(find /tmp/prof_w -name "*counter_collection.csv" | head -1) $R/gpurun_out/final/gemm_traffic.json; cat $R/gpurun_out/final/gemm_traffic.json | head -20
head -12 $R/gpurun_out/final/unet_bench_kernel_stats.csv | cut -c1-200
