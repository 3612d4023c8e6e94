#!/bin/bash
# round 6, call 34: 128 x 128 output tile in t2v_wgrad_tn for full fine-tuning's base-weight gradients: device tests, per-shape A/B by token split, step A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c34
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_unet_grad.py -q -x -m gpu -k "wgrad" 2>&1 | tail -3 | tee $O/pytest_wgrad.txt
for t in 0 1; do
  T2V_WGRAD_TILE128=$t timeout 600 python tools/wgrad_full_time.py --splits 0,2,4,8,16,32 > $O/wgrad_full_tile128_$t.csv 2> $O/wgrad_full_$t.err
  tail -2 $O/wgrad_full_tile128_$t.csv
done
for i in 1 2; do
  for t in 0 1; do
    T2V_WGRAD_TILE128=$t timeout 600 python tools/full_finetune_time.py --frames 16 --steps 4 2> $O/ff_$t.err | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'tile128': $t, 'step_ms': d['step_ms'], 'grad_norm': d.get('grad_norm'), 'finite': d.get('all_grads_finite')}))" | tee -a $O/full_finetune_tile128_ab.jsonl
  done
done
timeout 900 python -m pytest tests/test_gpu_train_parity.py -q -x -m gpu -k "full_fine" 2>&1 | tail -3 | tee $O/pytest_full.txt
