#!/bin/bash
# round 5, call 8: FULL sweep (every tile id x split, not only the round-2 candidate lists) of the inference step's GEMM shapes, then A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c8
mkdir -p $O
cd $R
timeout 1500 python tools/tune_gemm.py --vae 0 --widen 0 --train 0 --full 1 --out $O/gemm_tune_unet_full.json > $O/tune.log 2>&1; echo "tune rc=$?"; tail -2 $O/tune.log | cut -c1-300
python - $R/t2v-turbo_amd/gemm_tune.json $O/gemm_tune_unet_full.json $O/gemm_tune_merged.json <<'PY'
import json, sys
old = json.load(open(sys.argv[1])); new = json.load(open(sys.argv[2]))
k = lambda r: (r["mode"], r["M"], r["N"], r["K"], r["batch"])
m = {k(r): r for r in old}; m.update({k(r): r for r in new})
json.dump(sorted(m.values(), key=k), open(sys.argv[3], "w"), indent=0)
print("merged", len(m))
PY
for t in old new old new; do
  F=$R/t2v-turbo_amd/gemm_tune.json; [ $t = new ] && F=$O/gemm_tune_merged.json
  T2V_GEMM_TUNE_FILE=$F timeout 300 python bench.py --clip 0 --cpu-baseline 0 --breakdown 0 --distill 0 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'table':'$t','unet_ms':r['ms_per_step']}))"
done | tee $O/table_ab.jsonl
