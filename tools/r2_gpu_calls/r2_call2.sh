#!/bin/bash
# Round-2 GPU call 2: the v1 distillation step with the native student (variants), the re-run of the fixed dropout test, rocprof of the step.
set -u
mkdir -p gpurun_out/c2
timeout 120 python -m pytest tests/test_gpu_unet_grad.py -m gpu -q --tb=short -p no:cacheprovider -k dropout > gpurun_out/c2/dropout.txt 2>&1; tail -3 gpurun_out/c2/dropout.txt
timeout 600 python tools/distill_bench.py --steps 3 --native-variants plain,graph,flash,flash+tn,flash+tn+graph > gpurun_out/c2/native_variants.txt 2> gpurun_out/c2/native_variants.err
grep '^{' gpurun_out/c2/native_variants.txt; tail -5 gpurun_out/c2/native_variants.err
timeout 300 python tools/distill_bench.py --steps 3 --batch-teacher 1 --native-variants flash+tn+graph > gpurun_out/c2/native_batched_teacher.txt 2> gpurun_out/c2/native_batched_teacher.err
grep '^{' gpurun_out/c2/native_batched_teacher.txt; tail -3 gpurun_out/c2/native_batched_teacher.err
timeout 300 python tools/distill_bench.py --steps 3 > gpurun_out/c2/torch_student.txt 2> gpurun_out/c2/torch_student.err; tail -1 gpurun_out/c2/torch_student.txt
cd /tmp; export TMPDIR=/tmp
T2V_FLASH_ATTN_BWD=1 T2V_TN_WGRAD=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c2/prof -- \
    python $GRAFT_REPO_ROOT/tools/distill_bench.py --steps 2 --warmup 1 --native-student 1 > $GRAFT_REPO_ROOT/gpurun_out/c2/prof.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/c2/prof.log
find $GRAFT_REPO_ROOT/gpurun_out/c2/prof -name "*kernel_stats.csv" | head; find $GRAFT_REPO_ROOT/gpurun_out/c2/prof -name "*kernel_trace.csv" -delete
