#!/bin/bash
# round 6, call 17: GroupNorm in proj_in's panel fill (gn_in: t2v_gn_coef_cs + t2v_linear_pr with gn_coef): device tests, UNet step A/B (T2V_GN_IN=0/1),
# engine parity at full width; then the full fine-tuning student at full width (tools/full_finetune_time.py)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c17
mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "linear_pr or gn_coef" 2>&1 | tail -8 ) > $O/t_kernels.txt 2>&1
tail -3 $O/t_kernels.txt
for v in 0 1 0 1; do
  T2V_GN_IN=$v timeout 400 python bench.py --clip 0 --cpu-baseline 0 --distill 0 --breakdown 0 2>$O/bench_$v.err | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'gn_in':$v,'ms_per_step':r['ms_per_step'],'launches':r['config'].get('launches_per_step')}))"
done | tee $O/step_ab.jsonl
( timeout 1500 python -m pytest -q -m gpu "tests/test_gpu_engine.py::test_unet_full_width_c2_config_vs_oracle" "tests/test_gpu_engine.py::test_unet_tiny_vs_reference_golden" 2>&1 | tail -5 ) > $O/t_engine.txt 2>&1
tail -3 $O/t_engine.txt
timeout 900 python tools/full_finetune_time.py --frames 16 --steps 3 > $O/full_finetune.json 2> $O/full_finetune.err; tail -1 $O/full_finetune.json; tail -3 $O/full_finetune.err
