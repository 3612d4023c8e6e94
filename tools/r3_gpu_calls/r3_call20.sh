#!/bin/bash
# round 3, call 20-21: the bench distillation leg with its student seeded; (21) after use_graph = False really means plain replay
set -u
mkdir -p gpurun_out/r3c21
timeout 420 python bench.py --cpu-baseline 0 --clip 0 --steps 10 > gpurun_out/r3c21/bench.json 2> gpurun_out/r3c21/bench.err; echo "rc=$?"
python - <<'PY'
import json
r = json.loads([l for l in open('gpurun_out/r3c21/bench.json') if l.startswith('{"metric"')][-1])
d = r['distill_step']
print('unet', r['ms_per_step'], 'distill', d['ms_per_step'], d['issue'], d['ms_per_step_by_issue'])
print({k: d['parity'][k] for k in ('out_rel_l2', 'dx_rel_l2', 'lora_grad_cos_min', 'lora_grad_norm_ratio_max_dev', 'ok')})
PY
