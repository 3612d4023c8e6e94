#!/bin/bash
set -u
mkdir -p gpurun_out/c20
R=$GRAFT_REPO_ROOT
timeout 400 python tools/distill_bench.py --steps 4 --module-route 1 > gpurun_out/c20/module.txt 2> gpurun_out/c20/module.err; grep '^{' gpurun_out/c20/module.txt | cut -c1-300
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c20 -- python $R/tools/distill_bench.py --steps 2 --warmup 2 --module-route 1 > $R/gpurun_out/c20/prof.log 2>&1
python $R/tools/trace_window.py /tmp/prof_c20 --marker sinh --steps 2 --out $R/gpurun_out/c20/module_window_stats.csv
python - <<'PY'
import csv,os
rows=list(csv.reader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/c20/module_window_stats.csv')))[2:]
ours=('gemm_kernel','dropout','wgrad','splitk','attn_','gn_','gather_kernel','geglu','layernorm','transpose','softmax','scatter','sumpool','add_bf16','fill_zero','conv_small','cast_kernel','ncfhw','tokens_to','timestep_emb','adamw','sumsq','silu_kernel','lcm','lincomb','add_kernel')
o=a=0; na=0
for r in rows:
    ms=float(r[2])
    if any(k in r[0] for k in ours): o+=ms
    else:
        a+=ms; na+=float(r[1])
        if ms>0.3: print(f"{r[0][:110]:110s} {r[1]:>8} {ms:7.3f}")
print("native ms/step",o,"other ms/step",a,"other kernels/step",na)
PY
