#!/bin/bash
# round 4, call 18: what bounds t2v_wgrad_tn_group? three separate --pmc passes (kernel-trace only) over tools/wgrad_pmc_target.py
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${RUN:-r4c18}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/p_wg_$i
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/p_wg_$i -- python $R/tools/wgrad_pmc_target.py > /dev/null 2>$O/pass_$i.err
  python3 - "$i" /tmp/p_wg_$i/*/*counter_collection.csv >> $O/pmc.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[2])):
    n = r["Kernel_Name"]
    if "wgrad_tn_group" in n:
        k = ("reduce" if "reduce" in n else "main", r["Counter_Name"])
        a = acc[k]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"{k[0]},pass{sys.argv[1]},{k[1]},{acc[k][0] / acc[k][1]:.0f},{acc[k][1]}")
PY
done
cat $O/pmc.csv
