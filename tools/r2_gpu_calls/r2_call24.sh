#!/bin/bash
set -u
mkdir -p gpurun_out/c24
timeout 400 python tools/gemm_profile_graph.py --blas 1 --out gpurun_out/c24/gemm_shapes_ingraph.csv > gpurun_out/c24/g.log 2>&1; tail -1 gpurun_out/c24/g.log | cut -c1-200
timeout 300 python tools/op_profile_graph.py --out gpurun_out/c24/ops_ingraph.csv > gpurun_out/c24/o.log 2>&1; head -5 gpurun_out/c24/ops_ingraph.csv
