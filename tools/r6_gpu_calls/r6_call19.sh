#!/bin/bash
# round 6, call 19: t2v_linear_pr at K = 512 (the 8-head temporal transformer behind the entry conv: q|k|v and GEGLU with LayerNorm in the fill):
# device tests, UNet step A/B (T2V_LPR_K512=0/1), engine parity at full width
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c19
mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "linear_pr" 2>&1 | tail -8 ) > $O/t_kernels.txt 2>&1
tail -3 $O/t_kernels.txt
for v in 0 1 0 1; do
  T2V_LPR_K512=$v timeout 400 python bench.py --clip 0 --cpu-baseline 0 --distill 0 --breakdown 0 2>$O/bench_$v.err | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'lpr_k512':$v,'ms_per_step':r['ms_per_step'],'launches':r['config'].get('launches_per_step')}))"
done | tee $O/step_ab.jsonl
( timeout 1500 python -m pytest -q -m gpu "tests/test_gpu_engine.py::test_unet_full_width_c2_config_vs_oracle" "tests/test_gpu_engine.py::test_unet_tiny_vs_reference_golden" 2>&1 | tail -5 ) > $O/t_engine.txt 2>&1
tail -3 $O/t_engine.txt
