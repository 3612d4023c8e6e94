#!/bin/bash
# round 3, call 18: the full-width training parity gate and smoke() on the final library (the dropout-epilogue tile rule changed after call 15)
set -u
mkdir -p gpurun_out/r3c18
timeout 900 python -m pytest tests/test_gpu_train_parity.py::test_student_full_width_forward_backward_vs_cpu_autograd -m gpu -q -x -s > gpurun_out/r3c18/fullwidth.txt 2>&1
grep -E "full-width|\[.*\] [0-9]+ LoRA|passed|failed|out |dx " gpurun_out/r3c18/fullwidth.txt | tail -6 | cut -c1-400
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
