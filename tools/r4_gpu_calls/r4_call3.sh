#!/bin/bash
# round 4, call 3: t2v_conv_halo ablation (where does ITS launch time go) + PMC passes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c3
mkdir -p $O
cd $R
export T2V_LAB_LIBS=$R/t2v-turbo_amd/libt2v_hip.so:$R/t2v-turbo_amd/libt2v_hip_ablate.so
timeout 240 tools/gemm_lab tools/r4_gpu_calls/spec_halo_abl.txt > $O/abl.csv 2> $O/abl.err
cat $O/abl.csv
export TMPDIR=/tmp
cd /tmp
for f in pmc_h0 pmc_h1; do
  i=0
  for set in "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
             "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
             "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rm -rf /tmp/p_${f}_$i
    timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/p_${f}_$i -- $R/tools/gemm_lab $R/tools/r4_gpu_calls/spec_$f.txt > /dev/null 2>$O/${f}_$i.err
    python3 - "$f" "$i" /tmp/p_${f}_$i/*/*counter_collection.csv >> $O/pmc.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[3])):
    if "conv_halo_kernel" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"{sys.argv[1]},pass{sys.argv[2]},{k},{acc[k][0] / acc[k][1]:.0f},{acc[k][1]}")
PY
  done
done
cat $O/pmc.csv
