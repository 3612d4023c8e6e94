#!/bin/bash
# round 6, call 43: t2v_wgrad_tn's block order (the shorter tile-grid axis fastest, so that an XCD's resident workgroups cover a near-square patch): tests, per-shape A/B, step A/B, FETCH_SIZE
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c43
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_unet_grad.py -q -x -m gpu -k "wgrad" 2>&1 | tail -3 | tee $O/pytest_wgrad.txt
for t in 0 1; do
  T2V_WGRAD_YFAST=$t timeout 600 python tools/wgrad_full_time.py --splits 0 > $O/wgrad_full_yfast_$t.csv 2> $O/wgrad_full_$t.err
  tail -1 $O/wgrad_full_yfast_$t.csv
done
for i in 1 2; do
  for t in 0 1; do
    T2V_WGRAD_YFAST=$t timeout 600 python tools/full_finetune_time.py --frames 16 --steps 4 2> $O/ff_$t.err | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'yfast': $t, 'step_ms': d['step_ms'], 'grad_norm': d.get('grad_norm')}))" | tee -a $O/yfast_ab.jsonl
  done
done
export TMPDIR=/tmp
cd /tmp
args=""
for t in 0 1; do
  for ctrs in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf /tmp/pw
    T2V_WGRAD_YFAST=$t timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pw -- python $R/tools/wgrad_full_pmc_target.py > $O/pmc_$t.log 2>&1
    f=$(find /tmp/pw -name "*counter_collection.csv" | head -1)
    c=$(echo $ctrs | cut -c1-3)
    if [ -n "$f" ]; then cp $f $O/cc_${t}_$c.csv; args="$args yfast$t=$O/cc_${t}_$c.csv"; fi
  done
done
python $R/tools/pmc_table.py $args > $O/wgrad_yfast_pmc.csv
rm -f $O/cc_*.csv
grep "wgrad_tn_kernel" $O/wgrad_yfast_pmc.csv
