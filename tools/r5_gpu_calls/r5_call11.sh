#!/bin/bash
# round 5, call 11: counters on the flash attention forward's 2 560-token launch (VERDICT r4 "next" 7): three separate --pmc passes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c11
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" \
           "SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_WAVES"; do
  i=$((i+1))
  rm -rf /tmp/p_attn_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/p_attn_$i -- python $R/tools/attn_one.py --nimg 16 --seq 2560 --heads 5 --iters 5 > /dev/null 2>$O/pmc_$i.err
  python3 - "$i" $(find /tmp/p_attn_$i -name "*counter_collection.csv" | head -1) >> $O/pmc.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[2])):
    if "attn_spatial" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"attn2560,pass{sys.argv[1]},{k},{acc[k][0] / acc[k][1]:.0f},{acc[k][1]}")
PY
done
cat $O/pmc.csv
