#!/bin/bash
# round 5, call 13: counters on the normalisation kernels at their in-step shapes (VERDICT r4 "next" 3a): one timing pass, then
# separate --pmc passes (SQ issue / wait, instruction mix, L2, FETCH_SIZE, WRITE_SIZE).  No library change.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c13
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 python $R/tools/norm_pmc_target.py --iters 3 > $O/target.txt 2>&1
rm -rf /tmp/p_norm_t
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_norm_t -- python $R/tools/norm_pmc_target.py --iters 20 > /dev/null 2>$O/stats.err
cp $(find /tmp/p_norm_t -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 - $(find /tmp/p_norm_t -name "*kernel_trace.csv" | head -1) > $O/durations.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "gn_" in n or "layernorm" in n:
        key = (n.split("(")[0][-60:], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
        acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
print("kernel,grid_x,grid_y,launches,avg_us,min_us")
for k, v in sorted(acc.items()):
    print(f"{k[0]},{k[1]},{k[2]},{len(v)},{sum(v) / len(v):.2f},{min(v):.2f}")
PY
i=0
for set in "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" \
           "FETCH_SIZE" \
           "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/p_norm_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/p_norm_$i -- python $R/tools/norm_pmc_target.py --iters 4 > /dev/null 2>$O/pmc_$i.err
  python3 - "$i" $(find /tmp/p_norm_$i -name "*counter_collection.csv" | head -1) >> $O/pmc.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[2])):
    n = r["Kernel_Name"]
    if "gn_" in n or "layernorm" in n:
        short = "gn_partial_cs" if "gn_partial_cs" in n else "gn_apply" if "gn_apply" in n else "layernorm_rows" if "layernorm_rows" in n else "layernorm" if "layernorm" in n else n[:30]
        key = (short, r.get("Grid_Size", "?"), r["Counter_Name"])
        a = acc[key]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"{k[0]},grid{k[1]},pass{sys.argv[1]},{k[2]},{acc[k][0] / acc[k][1]:.0f},{acc[k][1]}")
PY
done
cat $O/target.txt; cat $O/durations.csv; cat $O/pmc.csv
