#!/bin/bash
# round 6, call 7: does a late start of the second wave of every SIMD pay now that the epilogue has no waits of its own?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c7
mkdir -p $O
cd $R
L=$R/t2v-turbo_amd
export T2V_LAB_LIBS=$L/libt2v_hip.so:$L/libt2v_hip_v1.so:$L/libt2v_hip_v2.so:$L/libt2v_hip_v3.so:$L/libt2v_hip_v4.so:$L/libt2v_hip_v5.so:$L/libt2v_hip_v6.so
timeout 600 tools/linear_lab tools/r6_gpu_calls/spec_lpr_c7.txt > $O/lab.csv 2> $O/lab.err
cut -d, -f1,12,13,15 $O/lab.csv
tail -5 $O/lab.err
T2V_LAB_LIBS=$L/libt2v_hip_tr0.so LPR_TRACE=1 timeout 100 tools/linear_lab tools/r6_gpu_calls/spec_lpr_trace.txt > $O/trace.txt 2>&1
grep "wg-slot 1" $O/trace.txt
