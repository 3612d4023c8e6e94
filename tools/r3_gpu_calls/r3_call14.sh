#!/bin/bash
# round 3, call 14 (closing): the default bench line on the final code, kernel window of one steady-state distillation step
# (rocprofv3 kernel trace cut between two marker kernels), rocprofv3 summary of the UNet step loop, whole GPU suite
set -u
O=gpurun_out/r3c14
mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    r = json.loads([l for l in open('gpurun_out/r3c14/bench_line.json') if l.startswith('{"metric"')][-1])
    d = r['distill_step']
    print('unet ms', r['ms_per_step'], 'frac', r['roofline']['frac'], 'clip', r['clip_4step']['ms'], 'v2 clip', r['clip_16step_v2']['ms'])
    print('distill ms', d['ms_per_step'], 'fwd', d['forward_ms'], 'bwd', d['backward_ms'], 'parity', d['parity']['ok'], d['parity']['lora_grad_cos_min'], 'launches', d['launches'])
    print({k: (v['launches'], v['ms']) for k, v in d['forward_kernel_ms'].items()})
    print({k: (v['launches'], v['ms']) for k, v in d['backward_kernel_ms'].items()})
except Exception as e:
    print('bench FAILED', e); print(open('gpurun_out/r3c14/bench.err').read()[-1500:])
PY
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_train_parity.py::test_student_full_width_forward_backward_vs_cpu_autograd > $O/gpu_suite.txt 2>&1
grep -E "passed|failed" $O/gpu_suite.txt | tail -2
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_d -- python $R/tools/distill_bench.py --steps 2 --warmup 2 --native-student 1 > $R/$O/distill_trace.log 2>&1
python $R/tools/trace_window.py /tmp/prof_d --marker sinh --steps 2 --out $R/$O/distill_step_kernel_window.csv > $R/$O/window.log 2>&1; head -24 $R/$O/distill_step_kernel_window.csv | cut -c1-200
grep '^{' $R/$O/distill_trace.log | cut -c1-300
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --steps 10 --warmup 2 --clip 0 --cpu-baseline 0 --distill 0 --graph 0 > $R/$O/prof_stats.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $R/$O/unet_bench_kernel_stats.csv
head -6 $R/$O/unet_bench_kernel_stats.csv | cut -c1-200
