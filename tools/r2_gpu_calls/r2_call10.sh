#!/bin/bash
set -u
mkdir -p gpurun_out/c10
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "group_norm or groupnorm" > gpurun_out/c10/gn.txt 2>&1; tail -12 gpurun_out/c10/gn.txt | cut -c1-400
timeout 300 python -m pytest tests/test_gpu_engine.py -m gpu -q --tb=short -p no:cacheprovider -k "unet_tiny or full_width" > gpurun_out/c10/engine.txt 2>&1; tail -4 gpurun_out/c10/engine.txt | cut -c1-300
T2V_GN_COOP=1 timeout 300 python tools/op_profile_graph.py --out gpurun_out/c10/ops_coop.csv > gpurun_out/c10/ops_coop.log 2>&1; head -4 gpurun_out/c10/ops_coop.csv
T2V_GN_COOP=0 timeout 300 python tools/op_profile_graph.py --out gpurun_out/c10/ops_3launch.csv > gpurun_out/c10/ops_3launch.log 2>&1; head -2 gpurun_out/c10/ops_3launch.csv
timeout 300 python bench.py --cpu-baseline 0 --distill 0 --clip 0 > gpurun_out/c10/bench.json 2> gpurun_out/c10/bench.err; python -c "
import json; r=json.loads(open('gpurun_out/c10/bench.json').read().strip().splitlines()[-1]); print(r['ms_per_step'], r['roofline']['frac'], {k:v for k,v in r['kernel_ms'].items() if k in ('t2v_group_norm','t2v_gemm','t2v_layernorm')})"
