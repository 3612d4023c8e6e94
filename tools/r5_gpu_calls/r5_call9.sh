#!/bin/bash
# round 5, call 9: the flash attention forward at FOUR waves per SIMD (1 600 workgroups on 1 024 slots = 1.56 rounds instead of 2.08 on 768):
# the low-register variant (K / V fragments read in halves) at 3 and at 4 waves per SIMD against the product build; correctness of the
# variants by the device tests of the kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c9
mkdir -p $O
cd $R
for i in 0 1 2 3 0 1 2 3; do
  for shape in "--nimg 16 --seq 2560 --heads 5" "--nimg 16 --seq 640 --heads 10" "--nimg 16 --seq 160 --heads 20" "--nimg 16 --seq 2560 --kv 77 --heads 5"; do
    T2V_HIP_LIB=$R/t2v-turbo_amd/libt2v_hip_attn$i.so timeout 120 python tools/attn_one.py $shape --iters 20 2>/dev/null | sed "s/^/variant $i: /"
  done
done | tee $O/attn_variants.txt
T2V_HIP_LIB=$R/t2v-turbo_amd/libt2v_hip_attn2.so timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attn_spatial" 2>&1 | tail -2
