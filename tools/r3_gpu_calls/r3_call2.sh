#!/bin/bash
# round 3, call 2: fused normalisation statistics (GN column stats from GEMM epilogues, LN folded into consumers): kernel tests,
# whole GPU suite, bench A/B of the two switches
set -u
mkdir -p gpurun_out/r3c2
timeout 900 python -m pytest tests/test_gpu_gemm_fuse.py -m gpu -q -x > gpurun_out/r3c2/fuse_tests.txt 2>&1; tail -5 gpurun_out/r3c2/fuse_tests.txt | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_train_parity.py -m gpu -q -s -k trainer_route > gpurun_out/r3c2/route.txt 2>&1; grep -E "^\[|passed|failed|Error" gpurun_out/r3c2/route.txt | cut -c1-300
for v in "1 1" "0 0" "1 0" "0 1" "1 1"; do set -- $v
  T2V_FUSE_GN=$1 T2V_FOLD_LN=$2 timeout 300 python bench.py --steps 40 --cpu-baseline 0 --distill 0 --clip 0 > gpurun_out/r3c2/b_$1$2.json 2> gpurun_out/r3c2/b_$1$2.err
  python - <<PY
import json
try:
    r=json.loads(open('gpurun_out/r3c2/b_$1$2.json').read().strip().splitlines()[-1])
    k=r['kernel_ms']
    print('gn=$1 ln=$2', r['ms_per_step'], 'frac', r['roofline']['frac'], 'launches', r['config']['launches_per_step'], {n:(k[n]['launches'],k[n]['ms']) for n in k if n in ('t2v_gemm','t2v_group_norm','t2v_group_norm_cs','t2v_layernorm')})
except Exception as e: print('gn=$1 ln=$2 FAILED', e); print(open('gpurun_out/r3c2/b_$1$2.err').read()[-1500:])
PY
done
timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_train_parity.py::test_student_full_width_forward_backward_vs_cpu_autograd > gpurun_out/r3c2/gpu_suite.txt 2>&1; tail -6 gpurun_out/r3c2/gpu_suite.txt | cut -c1-300
