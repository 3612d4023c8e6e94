#!/bin/bash
# round 5, call 12: 64 queries per wave with the two query sets' phases offset (attn_spatial_q64_kernel): timing against the product kernel on
# the step's shapes (checksums must agree: same arithmetic per query), the kernel's device tests with it forced, and the UNet step A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c12
mkdir -p $O
cd $R
for pass in 1 2; do
  for v in 0 2; do
    for shape in "--nimg 16 --seq 2560 --heads 5" "--nimg 16 --seq 640 --heads 10" "--nimg 16 --seq 160 --heads 20" "--nimg 16 --seq 2560 --kv 77 --heads 5"; do
      T2V_ATTN_Q64=$v timeout 120 python tools/attn_one.py $shape --iters 20 2>/dev/null | sed "s/^/Q64=$v: /"
    done
  done
done | tee $O/attn_q64.txt
T2V_ATTN_Q64=2 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attn_spatial" 2>&1 | tail -2
for v in 0 1 0 1; do
  T2V_ATTN_Q64=$v timeout 300 python bench.py --clip 0 --cpu-baseline 0 --distill 0 --breakdown 0 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'T2V_ATTN_Q64':$v,'ms_per_step':r['ms_per_step']}))"
done | tee $O/step_q64_ab.jsonl
