#!/bin/bash
# round 3, call 9: weight prefetch from the GroupNorm normalise pass: bench A/B (interleaved), then the whole GPU suite
set -u
mkdir -p gpurun_out/r3c9
for v in 1 0 1 0; do
  T2V_PREFETCH=$v timeout 300 python bench.py --steps 40 --cpu-baseline 0 --distill 0 --clip 0 > gpurun_out/r3c9/b_pf$v.json 2> gpurun_out/r3c9/b_pf$v.err
  python - <<PY
import json
try:
    r=json.loads(open('gpurun_out/r3c9/b_pf$v.json').read().strip().splitlines()[-1])
    k=r['kernel_ms']
    print('prefetch=$v', r['ms_per_step'], 'frac', r['roofline']['frac'], 'launches', r['config']['launches_per_step'], {n:(k[n]['launches'],k[n]['ms']) for n in k if n in ('t2v_gemm','t2v_group_norm_cs','t2v_group_norm','t2v_layernorm')})
except Exception as e: print('prefetch=$v FAILED', e); print(open('gpurun_out/r3c9/b_pf$v.err').read()[-1500:])
PY
done
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_train_parity.py::test_student_full_width_forward_backward_vs_cpu_autograd > gpurun_out/r3c9/gpu_suite.txt 2>&1; tail -5 gpurun_out/r3c9/gpu_suite.txt | cut -c1-300
