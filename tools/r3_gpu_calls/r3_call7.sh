#!/bin/bash
# round 3, call 7: fused feed-forward: 2 vs 3 column tiles per wave, pinned vs compiler-scheduled issue order
set -u
mkdir -p gpurun_out/r3c7
for lib in "" _NOPIN; do for ct in 3 2; do
  echo "== lib$lib CT=$ct"
  T2V_HIP_LIB=t2v-turbo_amd/libt2v_hip$lib.so T2V_FFN_CT=$ct timeout 300 python tools/fuse_ab.py 2>/dev/null | grep -E "40960x320,feed_forward" | tee -a gpurun_out/r3c7/ab_ff.csv
done; done
T2V_FFN_CT=2 timeout 300 python -m pytest tests/test_gpu_gemm_fuse.py -m gpu -q -k fused_feed_forward 2>&1 | tail -2
