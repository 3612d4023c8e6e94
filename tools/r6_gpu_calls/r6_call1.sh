#!/bin/bash
# round 6, call 1: first contact of the panel-resident linear kernel (csrc/linear_pr.hip): correctness against t2v_gemm and an fp64
# reference, timing against the tuned t2v_gemm tiles, ablations, counters on the GEGLU launch
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c1
mkdir -p $O
cd $R
export TMPDIR=/tmp
export T2V_LAB_LIBS=$R/t2v-turbo_amd/libt2v_hip.so:$R/t2v-turbo_amd/libt2v_hip_ablate.so
timeout 600 tools/linear_lab tools/r6_gpu_calls/spec_lpr_first.txt > $O/lab.csv 2> $O/lab.err
cat $O/lab.csv | cut -c1-200
tail -5 $O/lab.err
cd /tmp
i=0
for set in "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/p_lpr_$i
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/p_lpr_$i -- $R/tools/linear_lab $R/tools/r6_gpu_calls/spec_lpr_pmc.txt > /dev/null 2>$O/pmc_$i.err
  python3 - "$i" $(find /tmp/p_lpr_$i -name "*counter_collection.csv" | head -1) >> $O/pmc.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[2])):
    if "linear_pr" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"ff1_lpr,pass{sys.argv[1]},{k},{acc[k][0] / acc[k][1]:.0f},{acc[k][1]}")
PY
done
cat $O/pmc.csv
