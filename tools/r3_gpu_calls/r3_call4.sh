#!/bin/bash
# round 3, call 4: diagnostic builds of the LayerNorm fold (no statistics loads / no hoisted fold vectors) on the consumer shapes
set -u
mkdir -p gpurun_out/r3c4
for v in "" _NOSTAT _NOHOIST; do
  echo "== lib$v"
  T2V_HIP_LIB=t2v-turbo_amd/libt2v_hip$v.so timeout 300 python tools/fuse_ab.py 2>/dev/null | grep -E "qkv|cross_q|ff1|out_proj" | tee gpurun_out/r3c4/ab$v.csv
done
timeout 600 python -m pytest tests/test_gpu_gemm_fuse.py -m gpu -q -x -k "layernorm_fold" 2>&1 | grep -E "rel_l2|assert|Error|passed|failed" | head -20
