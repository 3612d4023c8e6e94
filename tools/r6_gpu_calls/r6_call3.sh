#!/bin/bash
# round 6, call 3: which build-time knobs of linear_pr pay (anti-phase start of the two waves of a SIMD, rotated chunk walk, deeper ring)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c3
mkdir -p $O
cd $R
L=$R/t2v-turbo_amd
export T2V_LAB_LIBS=$L/libt2v_hip.so:$L/libt2v_hip_v1.so:$L/libt2v_hip_v2.so:$L/libt2v_hip_v3.so:$L/libt2v_hip_v4.so:$L/libt2v_hip_v5.so:$L/libt2v_hip_v6.so:$L/libt2v_hip_abl.so
timeout 600 tools/linear_lab tools/r6_gpu_calls/spec_lpr_variants.txt > $O/lab.csv 2> $O/lab.err
cut -d, -f1,12,13,15 $O/lab.csv
tail -5 $O/lab.err
