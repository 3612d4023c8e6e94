#!/bin/bash
set -u
mkdir -p gpurun_out/c22
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "layernorm_second" > gpurun_out/c22/k.txt 2>&1; tail -5 gpurun_out/c22/k.txt | cut -c1-300
timeout 400 python -m pytest tests/test_gpu_engine.py tests/test_modelscope.py -m gpu -q --tb=short -p no:cacheprovider -k "unet_tiny or full_width or alternating or modelscope or pipeline" > gpurun_out/c22/e.txt 2>&1; tail -4 gpurun_out/c22/e.txt | cut -c1-300
for i in 1 2; do for f in 1 0; do
T2V_FUSE_LN=$f timeout 300 python bench.py --steps 40 --cpu-baseline 0 --distill 0 --clip 0 > gpurun_out/c22/b.json 2> gpurun_out/c22/b.err; python -c "
import json; r=json.loads(open('gpurun_out/c22/b.json').read().strip().splitlines()[-1]); print('fuse_ln=$f', r['ms_per_step'], r['roofline']['frac'], r['config']['launches_per_step'], {k:(v['launches'],v['ms']) for k,v in r['kernel_ms'].items() if k in ('t2v_gemm','t2v_layernorm')})"
done; done
