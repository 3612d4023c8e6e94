#!/bin/bash
# Round-3 measurement call: the default bench line, the rocprofv3 kernel summary of the same UNet step, the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only) behind roofline.traffic, and the same-box A/B of the round's switches.
set -u
mkdir -p gpurun_out/r3final
timeout 1200 python bench.py > gpurun_out/r3final/bench_line.json 2> gpurun_out/r3final/bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r3final/bench.err | cut -c1-400
for v in "1 1" "0 0" "1 0" "0 1" "1 1" "0 0"; do set -- $v
  T2V_FUSE_GN=$1 T2V_FOLD_LN=$2 timeout 300 python bench.py --steps 40 --cpu-baseline 0 --distill 0 --clip 0 > gpurun_out/r3final/ab_$1$2.json 2>/dev/null
  python - <<PY
import json
r=json.loads(open('gpurun_out/r3final/ab_$1$2.json').read().strip().splitlines()[-1]); k=r['kernel_ms']
print(json.dumps({"T2V_FUSE_GN": $1, "T2V_FOLD_LN": $2, "ms_per_step": r["ms_per_step"], "roofline_frac": r["roofline"]["frac"], "launches": r["config"]["launches_per_step"],
                  "ms": {n: (k[n]["launches"], k[n]["ms"]) for n in k if n.startswith(("t2v_gemm", "t2v_group_norm", "t2v_layernorm"))}}))
PY
done | tee gpurun_out/r3final/switch_ab.jsonl
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --steps 10 --warmup 2 --clip 0 --cpu-baseline 0 --distill 0 --graph 0 > $R/gpurun_out/r3final/prof_stats.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r3final/unet_bench_kernel_stats.csv
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -- python $R/bench.py --steps 1 --warmup 1 --graph 0 --clip 0 --cpu-baseline 0 --distill 0 > $R/gpurun_out/r3final/prof_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -- python $R/bench.py --steps 1 --warmup 1 --graph 0 --clip 0 --cpu-baseline 0 --distill 0 > $R/gpurun_out/r3final/prof_write.log 2>&1
python $R/tools/pmc_traffic.py $(find /tmp/prof_f -name "*counter_collection.csv" | head -1) $(find /tmp/prof_w -name "*counter_collection.csv" | head -1) $R/gpurun_out/r3final/gemm_traffic.json; head -c 600 $R/gpurun_out/r3final/gemm_traffic.json
head -14 $R/gpurun_out/r3final/unet_bench_kernel_stats.csv | cut -c1-220
