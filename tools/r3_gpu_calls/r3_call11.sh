#!/bin/bash
# round 3, call 11: overlapped gradient exchange on RCCL (one rank), dQ kernel occupancy A/B, the whole GPU suite, bench line
set -u
O=gpurun_out/r3c11
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train_parity.py -m gpu -q -x -s -k "overlapped or checkpointing" > $O/overlap.txt 2>&1
grep -E "\[overlap\]|\[checkpoint\]|passed|failed|Error" $O/overlap.txt | tail -6 | cut -c1-300
{
  timeout 300 python tools/attn_bwd_ab.py | grep -E "build|spatial"
  T2V_AB_TAG=dq_2waves T2V_HIP_LIB=gpurun_variants/libt2v_abwd_dq2.so timeout 300 python tools/attn_bwd_ab.py | grep spatial
  timeout 300 python tools/attn_bwd_ab.py | grep spatial
} > $O/attn_bwd_dq_ab.csv 2> $O/attn_bwd_dq_ab.err
cat $O/attn_bwd_dq_ab.csv
timeout 900 python tools/distill_bench.py --native-student 1 --steps 4 --warmup 2 --native-variants "flash+tn,flash+tn+nocs,flash+tn,flash+tn+nocs" > $O/distill_gn_ab.jsonl 2> $O/distill_gn_ab.err
grep '^{' $O/distill_gn_ab.jsonl | cut -c1-420
timeout 1800 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_train_parity.py::test_student_full_width_forward_backward_vs_cpu_autograd > $O/gpu_suite.txt 2>&1
tail -4 $O/gpu_suite.txt | cut -c1-300
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err
python - <<'PY'
import json
try:
    r = json.loads([l for l in open('gpurun_out/r3c11/bench_line.json') if l.startswith('{"metric"')][-1])
    d = r['distill_step']
    print('unet ms', r['ms_per_step'], 'frac', r['roofline']['frac'], 'distill ms', d['ms_per_step'], 'fwd', d['forward_ms'], 'bwd', d['backward_ms'], 'parity', d['parity']['ok'], d['parity']['lora_grad_cos_min'])
    print({k: v for k, v in d['backward_kernel_ms'].items() if 'attn' in k})
except Exception as e:
    print('bench FAILED', e); print(open('gpurun_out/r3c11/bench.err').read()[-1500:])
PY
