#!/bin/bash
# Round-2 GPU call 3: where do the NaNs of the native student step come from; experimental tile ids vs the tuned table.
set -u
mkdir -p gpurun_out/c3
timeout 500 python tools/distill_nan_probe.py --steps 8 > gpurun_out/c3/nan_probe.txt 2> gpurun_out/c3/nan_probe.err
grep -E "^phase|non-finite|student out|parameters" gpurun_out/c3/nan_probe.txt | head -50; tail -3 gpurun_out/c3/nan_probe.err
timeout 300 python tools/gemm_profile_graph.py --blas 0 --force-cfgs 24,25,26,27,28,29 --top 40 --out gpurun_out/c3/gemm_experimental_cfgs.csv > gpurun_out/c3/gemm_exp.log 2>&1
tail -3 gpurun_out/c3/gemm_exp.log
