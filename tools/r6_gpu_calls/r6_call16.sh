#!/bin/bash
# round 6, call 16: deeper operand rings (tile ids 34-39) on the recorded descriptors of the UNet step, in-graph, next to the tuned choice
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c16
mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "cfg" 2>&1 | tail -4 ) > $O/t_cfgs.txt 2>&1; tail -2 $O/t_cfgs.txt
timeout 900 python tools/gemm_profile_graph.py --blas 0 --force-cfgs 34,35,36,37,38,39 --top 45 --out $O/gemm_deep_rings.csv > $O/deep.log 2>&1
tail -1 $O/deep.log
cut -d, -f1,3-7,9,11,12,14,15,20-25 $O/gemm_deep_rings.csv | head -50
