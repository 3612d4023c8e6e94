#!/bin/bash
set -u
mkdir -p gpurun_out/c14
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "group_norm or groupnorm" > gpurun_out/c14/gn.txt 2>&1; tail -3 gpurun_out/c14/gn.txt | cut -c1-400
timeout 300 python tools/op_profile_graph.py --out gpurun_out/c14/ops.csv > gpurun_out/c14/ops.log 2>&1; head -1 gpurun_out/c14/ops.csv
grep "t2v_group_norm" gpurun_out/c14/ops.csv | head -8
timeout 300 python bench.py --cpu-baseline 0 --distill 0 --clip 0 > gpurun_out/c14/bench.json 2> gpurun_out/c14/bench.err; python -c "
import json; r=json.loads(open('gpurun_out/c14/bench.json').read().strip().splitlines()[-1]); print(r['ms_per_step'], r['roofline']['frac'], {k:v for k,v in r['kernel_ms'].items() if k in ('t2v_group_norm','t2v_gemm','t2v_layernorm')})"
T2V_GN_TWO_LAUNCH=0 timeout 300 python bench.py --cpu-baseline 0 --distill 0 --clip 0 > gpurun_out/c14/bench_3l.json 2> gpurun_out/c14/bench_3l.err; python -c "
import json; r=json.loads(open('gpurun_out/c14/bench_3l.json').read().strip().splitlines()[-1]); print(r['ms_per_step'], r['roofline']['frac'], {k:v for k,v in r['kernel_ms'].items() if k in ('t2v_group_norm','t2v_gemm','t2v_layernorm')})"
