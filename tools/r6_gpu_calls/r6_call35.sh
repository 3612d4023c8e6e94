#!/bin/bash
# round 6, call 35: t2v_wgrad_tn's 128 x 128 tile after the tail-chunk fix (call 34: every launch with R or C = 320 serialised its loads in the ragged path),
# t2v_norm_affine_grad with chunks-per-lane / rows-in-flight by width: device tests, per-shape A/B, step A/B, kernel summary of the step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c35
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_unet_grad.py tests/test_gpu_kernels.py -q -x -m gpu -k "wgrad or affine or im2col" 2>&1 | tail -3 | tee $O/pytest_kernels.txt
for t in 0 1; do
  T2V_WGRAD_TILE128=$t timeout 600 python tools/wgrad_full_time.py --splits 0,8,16,32 > $O/wgrad_full_tile128_$t.csv 2> $O/wgrad_full_$t.err
  tail -1 $O/wgrad_full_tile128_$t.csv
done
for i in 1 2; do
  for t in 0 1; do
    T2V_WGRAD_TILE128=$t timeout 600 python tools/full_finetune_time.py --frames 16 --steps 4 2> $O/ff_$t.err | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'tile128': $t, 'step_ms': d['step_ms'], 'grad_norm': d.get('grad_norm'), 'finite': d.get('all_grads_finite')}))" | tee -a $O/full_finetune_tile128_ab.jsonl
  done
done
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_ff
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ff -- python $R/tools/full_finetune_time.py --frames 16 --steps 4 > $O/ff_prof.log 2>&1
cp $(find /tmp/prof_ff -name "*kernel_stats.csv" | head -1) $O/full_finetune_kernel_stats.csv
head -12 $O/full_finetune_kernel_stats.csv | cut -c1-100,200-330
cd $R
timeout 900 python -m pytest tests/test_gpu_train_parity.py -q -x -m gpu -k "full_fine" 2>&1 | tail -3 | tee $O/pytest_full.txt
