#!/bin/bash
# round 6, call 23 (recipe of r5_call7.sh): the tile table was last re-timed in round 5; since then some launches moved to t2v_linear_pr and the inference step's launches carry fused-statistics epilogues (their own
# tile ids) and the training step's the LoRA epilogue.  Re-time the stored candidates of every shape on the CURRENT descriptors, then A/B the
# two tables on the UNet step and on the distillation step (same box, interleaved).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c23
mkdir -p $O
cd $R
timeout 1200 python tools/tune_gemm.py --vae 1 --widen 1 --train 1 --out $O/gemm_tune_new.json > $O/tune.log 2>&1; echo "tune rc=$?"; tail -3 $O/tune.log | cut -c1-300
cp gpurun_out/gemm_tune_candidates.json $O/ 2>/dev/null
for t in old new old new; do
  F=$R/t2v-turbo_amd/gemm_tune.json; [ $t = new ] && F=$O/gemm_tune_new.json
  T2V_GEMM_TUNE_FILE=$F timeout 400 python bench.py --clip 0 --cpu-baseline 0 --breakdown 0 --distill-parity 0 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); ds=r['distill_step']
print(json.dumps({'table':'$t','unet_ms':r['ms_per_step'],'distill_ms':ds['ms_per_step'],'forward_ms':ds.get('forward_ms'),'backward_ms':ds.get('backward_ms')}))"
done | tee $O/table_ab.jsonl
