#!/bin/bash
# round 4, call 1: where does a t2v_gemm launch's time go TODAY (ablation build) + fresh PMC passes on the verdict's three families
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c1
mkdir -p $O
cd $R
export T2V_LAB_LIBS=$R/t2v-turbo_amd/libt2v_hip.so:$R/t2v-turbo_amd/libt2v_hip_ablate.so
timeout 240 tools/gemm_lab tools/r4_gpu_calls/spec_ablate.txt > $O/ablate.csv 2> $O/ablate.err
tail -3 $O/ablate.csv
export TMPDIR=/tmp
cd /tmp
for f in pmc_a pmc_b pmc_c pmc_d; do
  i=0
  for set in "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
             "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
             "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
    i=$((i+1))
    rm -rf /tmp/p_${f}_$i
    timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/p_${f}_$i -- $R/tools/gemm_lab $R/tools/r4_gpu_calls/spec_$f.txt > /dev/null 2>$O/${f}_$i.err
    python3 - "$f" "$i" /tmp/p_${f}_$i/*/*counter_collection.csv >> $O/pmc.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[3])):
    if "gemm_kernel" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"{sys.argv[1]},pass{sys.argv[2]},{k},{acc[k][0] / acc[k][1]:.0f},{acc[k][1]}")
PY
  done
done
cat $O/pmc.csv | head -80
