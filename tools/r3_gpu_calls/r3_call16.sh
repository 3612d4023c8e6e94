#!/bin/bash
# round 3, call 16: bench line with the distillation leg timed under plain and hipGraph replay; training GPU tests on the final code
set -u
O=gpurun_out/r3c16
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_unet_grad.py tests/test_gpu_train_parity.py -m gpu -q -x \
   --deselect tests/test_gpu_train_parity.py::test_student_full_width_forward_backward_vs_cpu_autograd > $O/tests.txt 2>&1
grep -E "passed|failed" $O/tests.txt | tail -2
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    r = json.loads([l for l in open('gpurun_out/r3c16/bench_line.json') if l.startswith('{"metric"')][-1])
    d = r['distill_step']
    print('unet ms', r['ms_per_step'], 'frac', r['roofline']['frac'], 'clip', r['clip_4step']['ms'], 'v2 clip', r['clip_16step_v2']['ms'])
    print('distill ms', d['ms_per_step'], d['issue'], d['ms_per_step_by_issue'], 'fwd', d['forward_ms'], 'bwd', d['backward_ms'], 'parity', d['parity']['ok'], d['parity']['lora_grad_cos_min'])
    print('host', d['host_ms_last_step'])
except Exception as e:
    print('bench FAILED', e); print(open('gpurun_out/r3c16/bench.err').read()[-1500:])
PY
tail -5 $O/bench.err | cut -c1-200
