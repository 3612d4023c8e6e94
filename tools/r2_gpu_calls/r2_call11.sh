#!/bin/bash
set -u
mkdir -p gpurun_out/c11
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "group_norm or groupnorm" > gpurun_out/c11/gn.txt 2>&1; tail -5 gpurun_out/c11/gn.txt | cut -c1-400
T2V_GN_COOP=1 timeout 300 python tools/op_profile_graph.py --out gpurun_out/c11/ops_coop.csv > gpurun_out/c11/ops_coop.log 2>&1; head -1 gpurun_out/c11/ops_coop.csv
grep "t2v_group_norm" gpurun_out/c11/ops_coop.csv | head -12
timeout 300 python bench.py --cpu-baseline 0 --distill 0 --clip 0 > gpurun_out/c11/bench.json 2> gpurun_out/c11/bench.err; python -c "
import json; r=json.loads(open('gpurun_out/c11/bench.json').read().strip().splitlines()[-1]); print(r['ms_per_step'], r['roofline']['frac'], {k:v for k,v in r['kernel_ms'].items() if k in ('t2v_group_norm','t2v_gemm','t2v_layernorm')})"
T2V_GN_COOP=0 timeout 300 python bench.py --cpu-baseline 0 --distill 0 --clip 0 > gpurun_out/c11/bench_3l.json 2> gpurun_out/c11/bench_3l.err; python -c "
import json; r=json.loads(open('gpurun_out/c11/bench_3l.json').read().strip().splitlines()[-1]); print(r['ms_per_step'], r['roofline']['frac'], {k:v for k,v in r['kernel_ms'].items() if k in ('t2v_group_norm','t2v_gemm','t2v_layernorm')})"
