#!/bin/bash
# round 6, calls 10-11: t2v_linear_os (accumulators resident, activations streamed in 64-deep K slabs by two loader waves) against the tuned t2v_gemm tile
# (record only: t2v_linear_os and the LAB_OS switch of tools/linear_lab.cpp exist at commit 03c8032, not in the product)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c11
mkdir -p $O
cd $R
LAB_OS=1 timeout 300 tools/linear_lab tools/r6_gpu_calls/spec_los.txt > $O/lab.csv 2> $O/lab.err
cut -d, -f1,10,11,12,13,14,15 $O/lab.csv
tail -5 $O/lab.err
