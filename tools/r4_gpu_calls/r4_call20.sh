#!/bin/bash
# round 4, call 20: t2v_wgrad_tn_group with ds_read_b64_tr_b16 fragment reads (+ the XCD-contiguous block order): device tests, timing
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c20
mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_gpu_unet_grad.py -x -q -m gpu -k "wgrad" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 200 python tools/wgrad_time.py > $O/wgrad_tr.csv 2> $O/err.txt; cut -d, -f1,4,5,6,7 $O/wgrad_tr.csv
