#!/bin/bash
# round 3, call 12: grouped weight-gradient launches (kernel test, engine tests), distillation step A/B: grouped / per-product
# weight gradients, hipGraph replay of the student's lists (is the step host-bound on this box?)
set -u
O=gpurun_out/r3c12
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_unet_grad.py tests/test_gpu_train_parity.py tests/test_gpu_gemm_fuse.py -m gpu -q -x \
   --deselect tests/test_gpu_train_parity.py::test_student_full_width_forward_backward_vs_cpu_autograd > $O/tests.txt 2>&1
tail -3 $O/tests.txt | cut -c1-300
timeout 1200 python tools/distill_bench.py --native-student 1 --steps 6 --warmup 2 --native-variants "flash+tn,flash+tn+nogroup,flash+tn+graph,flash+tn,flash+tn+nogroup,flash+tn+graph" > $O/distill_ab.jsonl 2> $O/distill_ab.err
grep '^{' $O/distill_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['variant'], d['ms_per_step'], d['host_ms_last_step'])
"
tail -2 $O/distill_ab.err | cut -c1-300
