#!/bin/bash
# round 6, call 37: pack refresh of full fine-tuning: the data-gradient packs of the Linear leaves as native transposes of the refreshed forward packs; step time, kernel summary, parity tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${RUN:-r6c37}
mkdir -p $O
cd $R
for i in 1 2; do
  timeout 600 python tools/full_finetune_time.py --frames 16 --steps 4 2> $O/ff.err | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'step_ms': d['step_ms'], 'grad_norm': d.get('grad_norm'), 'finite': d.get('all_grads_finite'), 'peak_mem_gb': d.get('peak_mem_gb')}))" | tee -a $O/full_finetune_step.jsonl
done
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_ff
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ff -- python $R/tools/full_finetune_time.py --frames 16 --steps 4 > $O/ff_prof.log 2>&1
cp $(find /tmp/prof_ff -name "*kernel_stats.csv" | head -1) $O/full_finetune_kernel_stats.csv
cd $R
timeout 900 python -m pytest tests/test_gpu_train_parity.py -q -x -m gpu -k "full_fine" 2>&1 | tail -3 | tee $O/pytest_full.txt
