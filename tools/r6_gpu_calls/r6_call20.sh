#!/bin/bash
# round 6, call 20: the device tests of the T2V_EXPERIMENTAL entry points (skipped on the product library) on libt2v_hip_exp.so
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c20
mkdir -p $O
cd $R
export TMPDIR=/tmp
( T2V_HIP_LIB=$R/t2v-turbo_amd/libt2v_hip_exp.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gemm_fuse.py -q -m gpu -k "attn_spatial_forms or small_cout or fused_feed_forward" 2>&1 | tail -6 ) > $O/t_exp.txt 2>&1
tail -3 $O/t_exp.txt
