#!/bin/bash
# round 5, call 14: GroupNorm on column statistics, direct form (one block per group writes the per-channel affine, the apply pass
# starts on its rows at once) against the partial-sums form (T2V_GN_CS_DIRECT=0): kernel durations standalone, the UNet step A/B
# interleaved on one box, and the device tests that cover the op.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c14
mkdir -p $O
export TMPDIR=/tmp
cd $R
for d in 1 0; do
  T2V_GN_CS_DIRECT=$d timeout 200 python tools/norm_pmc_target.py --iters 3 --ln 2560:320 2>&1 | grep "^gn" | sed "s/^/direct=$d /"
done | tee $O/target.txt
cd /tmp
for d in 1 0; do
  rm -rf /tmp/p_gn_$d
  T2V_GN_CS_DIRECT=$d timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_gn_$d -- python $R/tools/norm_pmc_target.py --iters 20 --gn 320:16:2560,320:1:40960,640:1:10240,1280:1:2560,1280:16:160 --ln 2560:320 > /dev/null 2>$O/trace_$d.err
  python3 - $d $(find /tmp/p_gn_$d -name "*kernel_trace.csv" | head -1) <<'PY' | tee -a $O/durations.csv
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    n = r["Kernel_Name"]
    if "gn_" in n:
        short = "gn_coef_cs" if "gn_coef_cs" in n else "gn_partial_cs" if "gn_partial_cs" in n else "gn_apply" if "gn_apply" in n else n[:30]
        acc[(short, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
for k, v in sorted(acc.items()):
    v = sorted(v)
    print(f"direct={sys.argv[1]},{k[0]},{k[1]},{k[2]},{len(v)},avg {sum(v) / len(v):.2f},median {v[len(v) // 2]:.2f},min {v[0]:.2f}")
PY
done
cd $R
for i in 1 2 3; do
  for d in 1 0; do
    T2V_GN_CS_DIRECT=$d timeout 300 python bench.py --steps 40 --warmup 5 --clip 0 --cpu-baseline 0 --distill 0 --breakdown 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); print('direct=$d ms_per_step', j['ms_per_step'])"
  done
done | tee $O/step_ab.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -q -m gpu -k "norm or unet or fused" 2>&1 | tail -3
