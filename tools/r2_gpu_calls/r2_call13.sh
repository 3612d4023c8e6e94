#!/bin/bash
set -u
mkdir -p gpurun_out/c13
timeout 500 python tools/distill_bench.py --steps 4 --native-student 1 > gpurun_out/c13/distill.txt 2> gpurun_out/c13/distill.err; grep '^{' gpurun_out/c13/distill.txt | cut -c1-330
timeout 300 python bench.py --cpu-baseline 0 --distill 0 --clip 1 > gpurun_out/c13/bench.json 2> gpurun_out/c13/bench.err; python -c "
import json; r=json.loads(open('gpurun_out/c13/bench.json').read().strip().splitlines()[-1]); print(r['ms_per_step'], r['roofline']['frac'], r['clip_4step'], {k:v for k,v in r['kernel_ms'].items() if k in ('t2v_group_norm','t2v_gemm','t2v_layernorm')})"
timeout 400 python -m pytest tests/test_gpu_unet_grad.py -m gpu -q --tb=short -p no:cacheprovider -k "gn_bwd or lora_training or unet_grad_engine" > gpurun_out/c13/grad.txt 2>&1; tail -3 gpurun_out/c13/grad.txt | cut -c1-300
