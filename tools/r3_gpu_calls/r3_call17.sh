#!/bin/bash
# round 3, call 17 (closing): whole GPU suite on the final code (incl. the segmented-graph gradient exchange on RCCL), the default
# bench line, kernel window of one steady-state distillation step
set -u
O=gpurun_out/r3c17
mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_train_parity.py::test_student_full_width_forward_backward_vs_cpu_autograd > $O/gpu_suite.txt 2>&1
grep -E "passed|failed" $O/gpu_suite.txt | tail -2
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    r = json.loads([l for l in open('gpurun_out/r3c17/bench_line.json') if l.startswith('{"metric"')][-1])
    d = r['distill_step']
    print('unet ms', r['ms_per_step'], 'frac', r['roofline']['frac'], 'clip', r['clip_4step']['ms'], 'v2 clip', r['clip_16step_v2']['ms'])
    print('distill ms', d['ms_per_step'], d['issue'], d['ms_per_step_by_issue'], 'fwd', d['forward_ms'], 'bwd', d['backward_ms'], 'parity', d['parity']['ok'], d['parity']['lora_grad_cos_min'])
except Exception as e:
    print('bench FAILED', e); print(open('gpurun_out/r3c17/bench.err').read()[-1500:])
PY
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_d -- python $R/tools/distill_bench.py --steps 2 --warmup 2 --native-student 1 > $R/$O/distill_trace.log 2>&1
python $R/tools/trace_window.py /tmp/prof_d --marker sinh --steps 2 --out $R/$O/distill_step_kernel_window.csv > $R/$O/window.log 2>&1; head -12 $R/$O/distill_step_kernel_window.csv | cut -c1-200
