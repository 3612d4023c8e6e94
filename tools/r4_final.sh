#!/bin/bash
# Round-4 measurement call: the default bench line, the rocprofv3 kernel summary of the same UNet step, the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only) behind roofline.traffic, and the same-box A/B of the round's switches.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4final
mkdir -p $O
cd $R
timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err | cut -c1-300
for v in 1 0 1 0; do
  T2V_CONV_HALO=$v timeout 300 python bench.py --steps 40 --cpu-baseline 0 --distill 0 --clip 0 > $O/ab_halo$v.json 2>/dev/null
  python - $O/ab_halo$v.json $v <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=r['kernel_ms']
print(json.dumps({"T2V_CONV_HALO": int(sys.argv[2]), "ms_per_step": r["ms_per_step"], "roofline_frac": r["roofline"]["frac"], "launches": r["config"]["launches_per_step"],
                  "ms": {n: (k[n]["launches"], k[n]["ms"]) for n in k if n.startswith(("t2v_gemm", "t2v_conv_halo", "t2v_group_norm"))}}))
PY
done | tee $O/switch_ab.jsonl
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_stats /tmp/prof_f /tmp/prof_w
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --steps 10 --warmup 2 --clip 0 --cpu-baseline 0 --distill 0 --graph 0 > $O/prof_stats.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/unet_bench_kernel_stats.csv
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -- python $R/bench.py --steps 1 --warmup 1 --graph 0 --clip 0 --cpu-baseline 0 --distill 0 > $O/prof_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -- python $R/bench.py --steps 1 --warmup 1 --graph 0 --clip 0 --cpu-baseline 0 --distill 0 > $O/prof_write.log 2>&1
python $R/tools/pmc_traffic.py $(find /tmp/prof_f -name "*counter_collection.csv" | head -1) $(find /tmp/prof_w -name "*counter_collection.csv" | head -1) $O/gemm_traffic.json; head -c 600 $O/gemm_traffic.json
head -14 $O/unet_bench_kernel_stats.csv | cut -c1-220
