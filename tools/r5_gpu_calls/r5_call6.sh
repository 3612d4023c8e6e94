#!/bin/bash
# round 5, call 6: issue-order switches of the flash attention forward (s_setprio around the MFMA clusters, no scheduling fences, two waves
# per SIMD) on the step's three spatial shapes, each variant its own library (tools/build_attn_variants.sh)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c6
mkdir -p $O
cd $R
for i in 0 1 2 3 4 5 0 1 2 3 4 5; do
  for shape in "--nimg 16 --seq 2560 --heads 5" "--nimg 16 --seq 640 --heads 10" "--nimg 16 --seq 160 --heads 20" "--nimg 16 --seq 2560 --kv 77 --heads 5"; do
    T2V_HIP_LIB=$R/t2v-turbo_amd/libt2v_hip_attn$i.so timeout 120 python tools/attn_one.py $shape --iters 20 2>/dev/null | sed "s/^/variant $i: /"
  done
done | tee $O/attn_variants.txt
