#!/bin/bash
set -u
mkdir -p gpurun_out/c17
timeout 900 python tools/tune_gemm.py --cold 1 --vae 0 --out gpurun_out/c17/gemm_tune_cold.json > gpurun_out/c17/tune.log 2>&1; tail -2 gpurun_out/c17/tune.log | cut -c1-200
python - <<'PY'
import json
cold={(r['mode'],r['M'],r['N'],r['K'],r['batch']):r for r in json.load(open('gpurun_out/c17/gemm_tune_cold.json'))}
cur=json.load(open('t2v-turbo_amd/gemm_tune.json'))
n=0
for r in cur:
    k=(r['mode'],r['M'],r['N'],r['K'],r['batch'])
    if k in cold and (cold[k]['cfg'],cold[k]['split'])!=(r['cfg'],r['split']):
        n+=1; r['cfg'],r['split']=cold[k]['cfg'],cold[k]['split']
json.dump(cur,open('gpurun_out/c17/gemm_tune_merged.json','w'),indent=0)
print("entries changed by the cold tune:", n)
PY
for i in 1 2; do
for tf in t2v-turbo_amd/gemm_tune.json gpurun_out/c17/gemm_tune_merged.json; do
T2V_GEMM_TUNE_FILE=$tf timeout 300 python bench.py --steps 40 --cpu-baseline 0 --distill 0 --clip 0 > gpurun_out/c17/b.json 2> gpurun_out/c17/b.err; python -c "
import json; r=json.loads(open('gpurun_out/c17/b.json').read().strip().splitlines()[-1]); print('$tf', r['ms_per_step'], r['roofline']['frac'], r['kernel_ms']['t2v_gemm']['ms'])"
done; done
