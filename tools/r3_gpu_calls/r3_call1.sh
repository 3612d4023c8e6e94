#!/bin/bash
# round 3, call 1: the training-parity gate + C4 full-width parity + the new bench line
set -u
mkdir -p gpurun_out/r3c1
free -g | head -2 > gpurun_out/r3c1/host.txt; nproc >> gpurun_out/r3c1/host.txt
timeout 1500 python -m pytest tests/test_gpu_train_parity.py -m gpu -q -s > gpurun_out/r3c1/train_parity.txt 2>&1
grep -E "^\[|passed|failed|Error|error" gpurun_out/r3c1/train_parity.txt | cut -c1-400 | tail -30
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -q -s -k "motion_cond_config_c4" > gpurun_out/r3c1/c4.txt 2>&1
grep -E "parity|passed|failed" gpurun_out/r3c1/c4.txt | cut -c1-300
timeout 900 python bench.py > gpurun_out/r3c1/bench.json 2> gpurun_out/r3c1/bench.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r3c1/bench.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r3c1/bench.json').read().strip().splitlines()[-1])
print(r['ms_per_step'], r['roofline']['frac'], r.get('clip_4step'), r.get('clip_16step_v2'))
print({k:v for k,v in r.get('cpu_baseline',{}).items() if k!='sample'})
d=r.get('distill_step',{}); print({k:d.get(k) for k in ('ms_per_step','forward_ms','backward_ms','parity','error')})
PY
