#!/bin/bash
# round 3, call 22 (last seconds of the budget): the LoRA epilogue kernel's first contact with hardware, then a short bench on the
# library as committed (the GEMM descriptor grew by the lora_* fields)
set -u
mkdir -p gpurun_out/r3c22
timeout 60 python -m pytest tests/gpu_unvalidated_lora_epilogue.py -m gpu -q -x 2>&1 | tail -4 | cut -c1-300
timeout 150 python bench.py --cpu-baseline 0 --clip 0 --steps 10 > gpurun_out/r3c22/bench.json 2> gpurun_out/r3c22/bench.err; echo "rc=$?"
python - <<'PY'
import json
r = json.loads([l for l in open('gpurun_out/r3c22/bench.json') if l.startswith('{"metric"')][-1])
d = r['distill_step']
print('unet', r['ms_per_step'], 'distill', d['ms_per_step'], d['issue'], d['ms_per_step_by_issue'])
print({k: d['parity'][k] for k in ('out_rel_l2', 'dx_rel_l2', 'lora_grad_cos_min', 'ok')})
PY
