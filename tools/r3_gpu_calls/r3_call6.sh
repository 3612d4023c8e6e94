#!/bin/bash
# round 3, call 6: the fused feed-forward kernel on hardware: parity, per-launch A/B, bench A/B; kernel tests after the policy change
set -u
mkdir -p gpurun_out/r3c6
timeout 600 python -m pytest tests/test_gpu_gemm_fuse.py -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed|rel_l2" | head -20
timeout 300 python tools/fuse_ab.py 2>/dev/null | grep -E "feed_forward|ff1|layernorm" | tee gpurun_out/r3c6/ab_ff.csv
for v in 1 0 1; do
  T2V_FUSE_FF=$v timeout 300 python bench.py --steps 40 --cpu-baseline 0 --distill 0 --clip 0 > gpurun_out/r3c6/b_ff$v.json 2> gpurun_out/r3c6/b_ff$v.err
  python - <<PY
import json
try:
    r=json.loads(open('gpurun_out/r3c6/b_ff$v.json').read().strip().splitlines()[-1])
    k=r['kernel_ms']
    print('ff=$v', r['ms_per_step'], 'frac', r['roofline']['frac'], 'launches', r['config']['launches_per_step'], {n:(k[n]['launches'],k[n]['ms']) for n in k if n in ('t2v_gemm','t2v_ffn_fused','t2v_group_norm_cs','t2v_group_norm','t2v_layernorm')})
except Exception as e: print('ff=$v FAILED', e); print(open('gpurun_out/r3c6/b_ff$v.err').read()[-1500:])
PY
done
