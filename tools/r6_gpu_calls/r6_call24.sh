#!/bin/bash
# round 6, call 24: counters on the FINAL t2v_linear_pr (three --pmc passes, kernel-trace only) next to t2v_gemm's tuned tile on the same shapes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c24
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
export T2V_LAB_LIBS=$R/t2v-turbo_amd/libt2v_hip.so
: > $O/pmc.csv
i=0
for set in "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/p_lpr_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/p_lpr_$i -- $R/tools/linear_lab $R/tools/r6_gpu_calls/spec_lpr_pmc_final.txt > $O/lab_$i.csv 2>$O/pmc_$i.err
  python3 - "$i" $(find /tmp/p_lpr_$i -name "*counter_collection.csv" | head -1) >> $O/pmc.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[2])):
    n = r["Kernel_Name"]
    fam = "linear_pr" if "linear_pr" in n else ("gemm" if "gemm_kernel" in n else None)
    if fam is None:
        continue
    # grid size tells the three shapes apart (workgroups x threads)
    key = (fam, r.get("Grid_Size", "?"), r["Counter_Name"])
    a = acc[key]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (fam, grid, c) in sorted(acc):
    a = acc[(fam, grid, c)]
    print(f"{fam},grid{grid},pass{sys.argv[1]},{c},{a[0] / a[1]:.0f},{a[1]}")
PY
done
cat $O/lab_1.csv | cut -d, -f1,10-13 | head -6
wc -l $O/pmc.csv
