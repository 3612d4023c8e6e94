#!/bin/bash
# round 5, call 17: the 49 GroupNorms of a UNet step that cannot take producer statistics, as ONE launch with a workgroup per (group, unit)
# (gn_group_kernel) against the two-launch form (T2V_GN_GROUP=0): UNet step interleaved on one box, then the device tests of the op and
# of the UNet engine.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c17
mkdir -p $O
cd $R
ab() { T2V_GN_GROUP=$1 timeout 200 python bench.py --steps 40 --warmup 5 --clip 0 --cpu-baseline 0 --distill 0 --breakdown 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); print('group=$1 ms_per_step', j['ms_per_step'])"; }
for i in 1 2; do ab 0; ab 1; done | tee $O/step_ab.txt
timeout 200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -q -m gpu -k "norm or unet" 2>&1 | tail -3 | tee $O/tests.txt
