#!/bin/bash
set -u
mkdir -p gpurun_out/c6
timeout 300 python -m pytest tests/test_gpu_engine.py -m gpu -q -x --tb=short -p no:cacheprovider -k "ddim_inversion" > gpurun_out/c6/f4.txt 2>&1; tail -8 gpurun_out/c6/f4.txt | cut -c1-300
timeout 900 python bench.py > gpurun_out/c6/bench.json 2> gpurun_out/c6/bench.err; echo "bench rc=$?"; tail -12 gpurun_out/c6/bench.err | cut -c1-400
python - <<'PY'
import json
r=json.loads(open('gpurun_out/c6/bench.json').read().strip().splitlines()[-1])
print({k:r[k] for k in ('value','ms_per_step')}, r.get('roofline',{}).get('frac'), r.get('clip_4step'))
print(json.dumps(r.get('cpu_baseline'))[:600])
print(json.dumps(r.get('distill_step'))[:3000])
PY
