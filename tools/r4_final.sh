#!/bin/bash
# Round-4 measurement call: the default bench line, the rocprofv3 kernel summary of the same UNet step, the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only) behind roofline.traffic, and the same-box A/B of the batched teacher call.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4final
mkdir -p $O
cd $R
timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err | cut -c1-300
# the teacher's two forwards as one 2-clip call (distill leg only), same box
for v in 0 1 0 1; do
  T2V_BATCH_TEACHER=$v timeout 400 python bench.py --clip 0 --cpu-baseline 0 --breakdown 0 2>/dev/null | tail -1 > $O/ab_bt$v.json
  python - $O/ab_bt$v.json $v <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); ds=r["distill_step"]
print(json.dumps({"T2V_BATCH_TEACHER": int(sys.argv[2]), "unet_ms": r["ms_per_step"], "distill_ms": ds["ms_per_step"], "by_issue": ds["ms_per_step_by_issue"],
                  "forward_ms": ds.get("forward_ms"), "backward_ms": ds.get("backward_ms")}))
PY
done | tee $O/batch_teacher_ab.jsonl
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_stats /tmp/prof_f /tmp/prof_w
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --steps 10 --warmup 2 --clip 0 --cpu-baseline 0 --distill 0 --graph 0 > $O/prof_stats.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/unet_bench_kernel_stats.csv
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -- python $R/bench.py --steps 1 --warmup 1 --graph 0 --clip 0 --cpu-baseline 0 --distill 0 > $O/prof_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -- python $R/bench.py --steps 1 --warmup 1 --graph 0 --clip 0 --cpu-baseline 0 --distill 0 > $O/prof_write.log 2>&1
python $R/tools/pmc_traffic.py $(find /tmp/prof_f -name "*counter_collection.csv" | head -1) $(find /tmp/prof_w -name "*counter_collection.csv" | head -1) $O/gemm_traffic.json; head -c 600 $O/gemm_traffic.json
head -14 $O/unet_bench_kernel_stats.csv | cut -c1-220
