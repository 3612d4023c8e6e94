#!/bin/bash
# round 5, call 10: the flash attention forward is L2 -> LDS fill-bound (every 128-query workgroup streams all keys and values: 1.05 GB per
# 2 560-token launch = 5.2 TB/s).  256 queries per workgroup (eight waves) halve that stream: product build (3 waves per SIMD: one 8-wave
# workgroup per CU) and the low-register build (4 per SIMD: two), each with T2V_ATTN_NW=4 / 8, two passes; then the kernel's device tests at NW=8
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c10
mkdir -p $O
cd $R
for pass in 1 2; do
for lib in libt2v_hip.so libt2v_hip_attn0.so; do
  for nw in 4 8; do
    for shape in "--nimg 16 --seq 2560 --heads 5" "--nimg 16 --seq 640 --heads 10" "--nimg 16 --seq 160 --heads 20" "--nimg 16 --seq 2560 --kv 77 --heads 5"; do
      T2V_ATTN_NW=$nw T2V_HIP_LIB=$R/t2v-turbo_amd/$lib timeout 120 python tools/attn_one.py $shape --iters 20 2>/dev/null | sed "s/^/$lib NW=$nw: /"
    done
  done
done
done | tee $O/attn_nw.txt
T2V_ATTN_NW=8 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attn_spatial" 2>&1 | tail -2
T2V_ATTN_NW=8 T2V_HIP_LIB=$R/t2v-turbo_amd/libt2v_hip_attn0.so timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attn_spatial" 2>&1 | tail -2
