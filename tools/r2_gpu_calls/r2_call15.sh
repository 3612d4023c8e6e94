#!/bin/bash
set -u
mkdir -p gpurun_out/c15
timeout 600 python -m pytest tests/test_gpu_unet_grad.py -m gpu -q --tb=short -p no:cacheprovider -k "dropout or lora_training" > gpurun_out/c15/grad.txt 2>&1; tail -4 gpurun_out/c15/grad.txt | cut -c1-300
timeout 500 python tools/distill_bench.py --steps 4 --native-student 1 > gpurun_out/c15/distill.txt 2> gpurun_out/c15/distill.err; grep '^{' gpurun_out/c15/distill.txt | cut -c1-330
T2V_FUSE_DROPOUT=0 timeout 500 python tools/distill_bench.py --steps 4 --native-student 1 > gpurun_out/c15/distill_unfused.txt 2> gpurun_out/c15/distill_unfused.err; grep '^{' gpurun_out/c15/distill_unfused.txt | cut -c1-330
timeout 300 python bench.py --cpu-baseline 0 --distill 0 --clip 0 > gpurun_out/c15/bench.json 2> gpurun_out/c15/bench.err; python -c "
import json; r=json.loads(open('gpurun_out/c15/bench.json').read().strip().splitlines()[-1]); print(r['ms_per_step'], r['roofline']['frac'], {k:v for k,v in r['kernel_ms'].items() if k in ('t2v_group_norm','t2v_gemm','t2v_layernorm')})"
