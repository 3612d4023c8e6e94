#!/bin/bash
# round 4, call 19: t2v_wgrad_tn_group with the XCD-contiguous block order: device tests, timing, the TCC pass again
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c19
mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_gpu_unet_grad.py -x -q -m gpu -k "wgrad" > $O/tests.txt 2>&1; tail -1 $O/tests.txt
timeout 200 python tools/wgrad_time.py > $O/wgrad_xcd.csv 2> $O/err.txt; cut -d, -f1,4,5,6,7 $O/wgrad_xcd.csv
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/p_wg_x
timeout 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d /tmp/p_wg_x -- python $R/tools/wgrad_pmc_target.py > /dev/null 2>$O/pass.err
python3 - /tmp/p_wg_x/*/*counter_collection.csv > $O/pmc_after.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "wgrad_tn_group" in n:
        k = ("reduce" if "reduce" in n else "main", r["Counter_Name"])
        a = acc[k]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"{k[0]},after,{k[1]},{acc[k][0] / acc[k][1]:.0f},{acc[k][1]}")
PY
cat $O/pmc_after.csv
