#!/bin/bash
# round 6, call 12: LayerNorm in t2v_linear_pr's panel fill (ln_in): device tests of the kernel, UNet step A/B (T2V_LN_IN=0/1), engine parity at full width
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c12
mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "linear_pr" 2>&1 | tail -8 ) > $O/t_kernels.txt 2>&1
tail -4 $O/t_kernels.txt
for v in 0 1 0 1; do
  T2V_LN_IN=$v timeout 400 python bench.py --clip 0 --cpu-baseline 0 --distill 0 --breakdown 0 2>$O/bench_$v.err | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'ln_in':$v,'ms_per_step':r['ms_per_step'],'parity':r.get('parity')}))"
done | tee $O/step_ab.jsonl
( timeout 1500 python -m pytest -q -m gpu "tests/test_gpu_engine.py::test_unet_full_width_c2_config_vs_oracle" "tests/test_gpu_engine.py::test_unet_tiny_vs_reference_golden" 2>&1 | tail -5 ) > $O/t_engine.txt 2>&1
tail -3 $O/t_engine.txt
