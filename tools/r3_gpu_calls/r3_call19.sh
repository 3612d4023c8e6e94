#!/bin/bash
# round 3, call 19: the full-width training parity gate with seeded weights on the final library (P / dS as hi + lo bf16 parts in the
# temporal attention backward), the temporal attention backward's kernel test and time
set -u
mkdir -p gpurun_out/r3c19
timeout 300 python -m pytest tests/test_gpu_unet_grad.py -m gpu -q -x -k "attn_temporal_bwd" 2>&1 | tail -2
timeout 200 python tools/attn_bwd_ab.py 2>/dev/null | grep temporal
timeout 900 python -m pytest tests/test_gpu_train_parity.py::test_student_full_width_forward_backward_vs_cpu_autograd -m gpu -q -x -s > gpurun_out/r3c19/fullwidth.txt 2>&1
grep -E "\[full width\]|passed|failed" gpurun_out/r3c19/fullwidth.txt | tail -4 | cut -c1-330
