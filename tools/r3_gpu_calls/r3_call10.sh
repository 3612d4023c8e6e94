#!/bin/bash
# round 3, call 10: attention backward kernels (temporal on the matrix cores, double-buffered spatial) A/B per launch,
# their GPU tests, native checkpointing on the device, and the distillation step with / without checkpointing
set -u
O=gpurun_out/r3c10
mkdir -p $O
{
  timeout 300 python tools/attn_bwd_ab.py
  T2V_TATTN_BWD_VALU=1 timeout 300 python tools/attn_bwd_ab.py | grep temporal
  T2V_AB_TAG=single_buffer T2V_HIP_LIB=gpurun_variants/libt2v_abwd_single.so timeout 300 python tools/attn_bwd_ab.py | grep spatial
  T2V_AB_TAG=dkv_2waves T2V_HIP_LIB=gpurun_variants/libt2v_abwd_wpe2.so timeout 300 python tools/attn_bwd_ab.py | grep spatial
  timeout 300 python tools/attn_bwd_ab.py | grep -v build,
} > $O/attn_bwd_ab.csv 2> $O/attn_bwd_ab.err
cat $O/attn_bwd_ab.csv
timeout 900 python -m pytest tests/test_gpu_unet_grad.py tests/test_gpu_train_parity.py -m gpu -q -x -s \
   --deselect tests/test_gpu_train_parity.py::test_student_full_width_forward_backward_vs_cpu_autograd > $O/tests.txt 2>&1
grep -E "checkpoint\]|passed|failed|Error|error" $O/tests.txt | tail -8 | cut -c1-400
timeout 900 python tools/distill_bench.py --native-student 1 --steps 4 --warmup 2 --native-variants "flash+tn,flash+tn+ckpt,flash+tn" > $O/distill_variants.jsonl 2> $O/distill_variants.err
cut -c1-700 $O/distill_variants.jsonl
timeout 600 python bench.py --steps 10 --cpu-baseline 0 --clip 0 --distill-parity 0 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
try:
    r = json.loads([l for l in open('gpurun_out/r3c10/bench.json') if l.startswith('{"metric"')][-1])
    d = r['distill_step']
    print('unet ms', r['ms_per_step'], 'distill ms', d['ms_per_step'], 'fwd', d['forward_ms'], 'bwd', d['backward_ms'])
    print({k: v for k, v in d['backward_kernel_ms'].items() if 'attn' in k})
except Exception as e:
    print('bench FAILED', e); print(open('gpurun_out/r3c10/bench.err').read()[-1500:])
PY
