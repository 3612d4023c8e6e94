#!/bin/bash
# round 4, call 5: halo kernel in the engine: kernel tests, full-width parity, UNet step A/B (T2V_CONV_HALO=0/1)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c5
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv_halo" 2>&1 | tail -5 | tee $O/halo_tests.txt
timeout 900 python -m pytest tests/test_gpu_engine.py -q -x -k "full_width or tiny_vs_reference" 2>&1 | tail -5 | tee $O/engine_tests.txt
for h in 1 0 1 0; do
  T2V_CONV_HALO=$h timeout 600 python bench.py --steps 20 --warmup 3 --clip 0 --distill 0 --cpu-baseline 0 2>$O/bench_h$h.err | tail -1 > $O/bench_h${h}_$RANDOM.json
done
for f in $O/bench_h*.json; do echo $f; python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read())
print({k:d[k] for k in ("value","ms_per_step") if k in d}, d.get("roofline"), d.get("parity"))
PY
done
