#!/bin/bash
set -u
mkdir -p gpurun_out/c18
for lib in libt2v_hip.so libt2v_hip_wpe2.so libt2v_hip_wpe4.so; do
for shape in "--nimg 16 --seq 2560 --heads 5" "--nimg 16 --seq 640 --heads 10" "--nimg 16 --seq 160 --heads 20" "--nimg 16 --seq 2560 --kv 77 --heads 5"; do
echo -n "$lib $shape: "; T2V_HIP_LIB=t2v-turbo_amd/$lib timeout 120 python tools/attn_one.py $shape --iters 20 2>&1 | tail -1
done; done
