#!/bin/bash
# PMC counters of the dominant kernel on the two shapes DESIGN.md section 8 talks about: the GEGLU projection of the 320-channel
# level (short K, 3200 workgroups) and the N = K = 320 out-projection with residual (one round of 256 workgroups).
set -u
mkdir -p gpurun_out/c31
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
for shape in "ff1 --mode 0 --m 40960 --cin 320 --n 2560 --act 1 --cfg 19" "outproj --mode 0 --m 40960 --cin 320 --n 320 --res 1 --cfg 23"; do
  set -- $shape; tag=$1; shift
  ARGS="$* --iters 10 --graph 0"
  echo "# $tag: tools/gemm_one.py $ARGS" >> $R/gpurun_out/c31/pmc.csv
  python $R/tools/gemm_one.py $* --iters 20 2>&1 | tail -1 | sed 's/^/# in-graph: /' >> $R/gpurun_out/c31/pmc.csv
  i=0
  for cs in "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
            "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
            "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1)); rm -rf /tmp/pmc_$i
    timeout 200 rocprofv3 --kernel-trace --pmc $cs --output-format csv -d /tmp/pmc_$i -- python $R/tools/gemm_one.py $ARGS > /dev/null 2>&1
    python - "$tag" "$i" /tmp/pmc_$i/*/*counter_collection.csv >> $R/gpurun_out/c31/pmc.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[3])):
    if "gemm_kernel" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"{sys.argv[1]},pmc{sys.argv[2]},{k},{acc[k][0] / acc[k][1]:.0f}")
PY
  done
done
cat $R/gpurun_out/c31/pmc.csv
