#!/bin/bash
set -u
mkdir -p gpurun_out/c16
for i in 1 2; do
for lib in libt2v_hip_prev.so libt2v_hip.so; do
T2V_HIP_LIB=t2v-turbo_amd/$lib timeout 300 python bench.py --steps 40 --cpu-baseline 0 --distill 0 --clip 0 > gpurun_out/c16/bench_$lib.$i.json 2> gpurun_out/c16/bench_$lib.$i.err; python -c "
import json,sys; r=json.loads(open('gpurun_out/c16/bench_$lib.$i.json').read().strip().splitlines()[-1]); print('$lib', r['ms_per_step'], r['roofline']['frac'], r['kernel_ms']['t2v_gemm']['ms'])"
done; done
