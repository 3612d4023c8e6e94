#!/bin/bash
set -u
mkdir -p gpurun_out/c30
timeout 900 python tools/tune_gemm.py --full 1 --vae 0 --out gpurun_out/c30/gemm_tune_full.json > gpurun_out/c30/tune.log 2>&1; tail -2 gpurun_out/c30/tune.log | cut -c1-200
python - <<'PY'
import json
new={(r['mode'],r['M'],r['N'],r['K'],r['batch']):r for r in json.load(open('gpurun_out/c30/gemm_tune_full.json'))}
cur=json.load(open('t2v-turbo_amd/gemm_tune.json'))
n=0; gain=0
for r in cur:
    k=(r['mode'],r['M'],r['N'],r['K'],r['batch'])
    if k in new and (new[k]['cfg'],new[k]['split'])!=(r['cfg'],r['split']):
        n+=1; print(k, (r['cfg'],r['split']), '->', (new[k]['cfg'],new[k]['split']), r['us'], new[k]['us'], 'x', new[k]['count'])
        r['cfg'],r['split']=new[k]['cfg'],new[k]['split']
json.dump(cur,open('gpurun_out/c30/gemm_tune_merged.json','w'),indent=0)
print("changed", n)
PY
for i in 1 2; do for tf in t2v-turbo_amd/gemm_tune.json gpurun_out/c30/gemm_tune_merged.json; do
T2V_GEMM_TUNE_FILE=$tf timeout 300 python bench.py --steps 40 --cpu-baseline 0 --distill 0 --clip 0 > gpurun_out/c30/b.json 2> gpurun_out/c30/b.err; python -c "
import json; r=json.loads(open('gpurun_out/c30/b.json').read().strip().splitlines()[-1]); print('$tf', r['ms_per_step'], r['roofline']['frac'])"
done; done
