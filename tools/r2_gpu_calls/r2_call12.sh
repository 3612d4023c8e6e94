#!/bin/bash
set -u
mkdir -p gpurun_out/c12
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "group_norm or groupnorm" > gpurun_out/c12/gn.txt 2>&1; tail -3 gpurun_out/c12/gn.txt | cut -c1-400
T2V_GN_COOP=1 timeout 300 python tools/op_profile_graph.py --out gpurun_out/c12/ops_coop.csv > gpurun_out/c12/ops_coop.log 2>&1; head -1 gpurun_out/c12/ops_coop.csv
grep "t2v_group_norm" gpurun_out/c12/ops_coop.csv | head -8
timeout 1000 python tools/tune_gemm.py --train 1 --out gpurun_out/c12/gemm_tune.json > gpurun_out/c12/tune.log 2>&1; tail -3 gpurun_out/c12/tune.log | cut -c1-300
ls -la gpurun_out/c12/ gpurun_out/*.json 2>/dev/null | head
