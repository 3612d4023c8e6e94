#!/bin/bash
# round 4, call 4: t2v_conv_halo v2 (raw-buffer LDS-DMA, constant-stride fragment addresses): check + time + ablation
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c4
mkdir -p $O
cd $R
export T2V_LAB_LIBS=$R/t2v-turbo_amd/libt2v_hip.so:$R/t2v-turbo_amd/libt2v_hip_ablate.so
timeout 300 tools/gemm_lab tools/r4_gpu_calls/spec_halo2.txt > $O/halo2.csv 2> $O/halo2.err
cat $O/halo2.csv
tail -5 $O/halo2.err
