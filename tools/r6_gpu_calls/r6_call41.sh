#!/bin/bash
# round 6, call 41: pack refresh in two parts (the backward-only packs issued behind the forward's launches) + fp32 packs that ARE the parameter skipped: step A/B on one box, parity tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c41
mkdir -p $O
cd $R
for i in 1 2; do
  for t in 0 1; do
    T2V_REFRESH_SPLIT=$t timeout 600 python tools/full_finetune_time.py --frames 16 --steps 4 2> $O/ff_$t.err | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'refresh_split': $t, 'step_ms': d['step_ms'], 'pack_refresh_ms': d.get('pack_refresh_ms'), 'grad_norm': d.get('grad_norm')}))" | tee -a $O/refresh_split_ab.jsonl
  done
done
timeout 900 python -m pytest tests/test_gpu_train_parity.py -q -x -m gpu -k "full_fine" 2>&1 | tail -3 | tee $O/pytest_full.txt
