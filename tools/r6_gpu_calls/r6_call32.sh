#!/bin/bash
# round 6, call 32: large weight-gradient products of full fine-tuning on transposed copies + K-contiguous t2v_gemm (T2V_FULL_GEMM_WGRAD=1) against
# t2v_wgrad_tn everywhere (=0): the full-width parity test both ways, the step time both ways
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c32
mkdir -p $O
cd $R
export TMPDIR=/tmp
for v in 0 1 0 1; do
  T2V_FULL_GEMM_WGRAD=$v timeout 900 python tools/full_finetune_time.py --frames 16 --steps 4 2>$O/ff_$v.err | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'gemm_wgrad':$v,'step_ms':j['step_ms'],'grad_norm':j['grad_norm'],'finite':j['all_grads_finite'],'launches':j['launches']}))"
done | tee $O/ab.jsonl
( T2V_FULL_GEMM_WGRAD=1 timeout 1500 python -m pytest tests/test_gpu_train_parity.py -q -m gpu -s -k "full_fine_tuning" 2>&1 | grep -v "^$" | tail -8 ) > $O/t_full.txt 2>&1; tail -5 $O/t_full.txt | cut -c1-300
