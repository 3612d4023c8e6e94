#!/bin/bash
# SQ / TCC counters of the GEMM kernel on one representative conv shape, three separate --pmc passes (rocprofv3 cannot
# mix them in one run: 8 SQ slots, 4 TCC slots).  Usage (on an MI355X): bash tools/pmc_gemm_one.sh > profiles/rNN_gemm_pmc_counters.csv
R=$(cd "$(dirname "$0")/.." && pwd)
ARGS="--mode 1 --nimg 16 --h 20 --w 32 --cin 640 --n 640 --cfg ${CFG:-7} --iters 10 --graph 0"
export TMPDIR=/tmp
cd /tmp
echo "# rocprofv3 --pmc passes on conv3x3 M=10240 N=640 K=5760 (tools/gemm_one.py $ARGS)"
echo "# values are per dispatch (mean over the gemm_kernel dispatches); three separate passes (SQ x2, TCC)"
echo "pass,counter,value"
i=0
for set in "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
           "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$i -- python $R/tools/gemm_one.py $ARGS > /dev/null 2>&1
  python - "$i" /tmp/pmc_$i/*/*counter_collection.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[2])):
    if "gemm_kernel" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"pmc{sys.argv[1]},{k},{acc[k][0] / acc[k][1]:.0f}")
PY
done
