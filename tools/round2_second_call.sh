#!/bin/bash
# Second GPU call of round 2 (after round2_first_call.sh is green; ~12 GPU-minutes): the v1 distillation step with the native student.
#   1. torch student (the 728 ms of round 1) for the same-box reference
#   2. native student, five engine variants on one model build: plain replay, hipGraph, flash attention backward, + token-contracted
#      weight gradients, + both with graphs
#   3. rocprofv3 kernel summary of the plain variant (copy the *_kernel_stats.csv you want judged into profiles/)
# then: python tools/tune_gemm.py --train 1    (the student step's 359 GEMM shapes join the tuned table)
# usage: gpurun --timeout 1200 -- bash tools/round2_second_call.sh
set -u
mkdir -p gpurun_out
timeout 300 python tools/distill_bench.py --steps 3 2>&1 | tail -1 | tee gpurun_out/distill_torch_student.txt
timeout 600 python tools/distill_bench.py --steps 3 --native-variants plain,graph,flash,flash+tn,flash+tn+graph 2>&1 \
    | grep '^{' | tee gpurun_out/distill_native_variants.txt
timeout 400 python tools/distill_bench.py --steps 3 --batch-teacher 1 --native-variants flash+tn+graph 2>&1 \
    | grep '^{' | tee gpurun_out/distill_native_batched_teacher.txt
export TMPDIR=/tmp
T2V_FLASH_ATTN_BWD=1 T2V_TN_WGRAD=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_distill_native -- \
    python tools/distill_bench.py --steps 2 --warmup 1 --native-student 1 > gpurun_out/prof_distill_native.log 2>&1
