#!/bin/bash
# round 6, call 40: is the pack refresh host-bound?  refresh alone (host issue time / until the device is done) with the library repack kernel and with torch's chain, step A/B on one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c40
mkdir -p $O
cd $R
for i in 1 2; do
  for t in 0 1; do
    T2V_REPACK_NATIVE=$t timeout 600 python tools/full_finetune_time.py --frames 16 --steps 4 2> $O/ff_$t.err | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'repack_native': $t, 'step_ms': d['step_ms'], 'pack_refresh_ms': d.get('pack_refresh_ms'), 'grad_norm': d.get('grad_norm')}))" | tee -a $O/repack_ab.jsonl
  done
done
