#!/bin/bash
# round 6, call 44: per-shape / per-op in-graph tables of the UNet step on the FINAL tree (the committed ones are from before gn_coef and the K = 512 geometry)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c44
mkdir -p $O
cd $R
timeout 600 python tools/gemm_profile_graph.py --blas 0 --out $O/gemm_shapes_ingraph.csv > $O/gemm_shapes.log 2>&1; tail -1 $O/gemm_shapes.log
timeout 600 python tools/op_profile_graph.py --out $O/ops_ingraph.csv > $O/ops.log 2>&1; tail -2 $O/ops.log
head -1 $O/gemm_shapes_ingraph.csv
