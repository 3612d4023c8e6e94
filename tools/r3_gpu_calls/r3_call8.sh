#!/bin/bash
# round 3, call 8: SQ counters of the fused feed-forward kernel (why is it at 2x its estimate): two --pmc passes per variant
set -u
R=$(pwd)
mkdir -p gpurun_out/r3c8
export TMPDIR=/tmp
cd /tmp
for ct in 3 2; do
  export T2V_FFN_CT=$ct
  python $R/tools/ffn_one.py --iters 10 | tee -a $R/gpurun_out/r3c8/ffn_pmc.csv
  i=0
  for set in "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
             "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
             "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1)); rm -rf /tmp/pmc_$i
    timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$i -- python $R/tools/ffn_one.py --iters 5 > /dev/null 2>&1
    python - "$i" "$ct" /tmp/pmc_$i/*/*counter_collection.csv <<'PY' | tee -a $R/gpurun_out/r3c8/ffn_pmc.csv
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[3])):
    if "ffn_fused_kernel" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"ct{sys.argv[2]},pmc{sys.argv[1]},{k},{acc[k][0] / acc[k][1]:.0f}")
PY
  done
done
