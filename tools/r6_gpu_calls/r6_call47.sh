#!/bin/bash
# round 6, call 47: the pack refresh of full fine-tuning captured as one hipGraph (T2V_REFRESH_GRAPH=0/1): step A/B on one box, gradient norm after four weight updates, parity tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c47
mkdir -p $O
cd $R
for i in 1 2; do
  for t in 0 1; do
    T2V_REFRESH_GRAPH=$t timeout 600 python tools/full_finetune_time.py --frames 16 --steps 5 2> $O/ff_$t.err | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'refresh_graph': $t, 'step_ms': d['step_ms'], 'pack_refresh_ms': d.get('pack_refresh_ms'), 'grad_norm': d.get('grad_norm'), 'grad_norm_last_step': d.get('grad_norm_last_step'), 'peak_mem_gb': d.get('peak_mem_gb')}))" | tee -a $O/refresh_graph_ab.jsonl
  done
done
grep -i "warn\|error" $O/ff_1.err | head -5
timeout 900 python -m pytest tests/test_gpu_train_parity.py -q -x -m gpu -k "full_fine" 2>&1 | tail -3 | tee $O/pytest_full.txt
