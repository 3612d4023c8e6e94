#!/bin/bash
# round 4, call 16: token-contracted weight gradients with larger output tiles: device tests, per-group timing old / new, distillation step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c16
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_unet_grad.py -x -q -m gpu -k "wgrad" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
T2V_WGRAD_TILES=0 timeout 200 python tools/wgrad_time.py > $O/wgrad_64.csv 2> $O/err64.txt
timeout 200 python tools/wgrad_time.py > $O/wgrad_new.csv 2> $O/errnew.txt
T2V_WGRAD_BLOCKS=256 timeout 200 python tools/wgrad_time.py > $O/wgrad_new_256.csv 2> $O/errnew2.txt
T2V_WGRAD_BLOCKS=512 timeout 200 python tools/wgrad_time.py > $O/wgrad_new_512.csv 2> $O/errnew3.txt
paste -d'|' $O/wgrad_64.csv $O/wgrad_new.csv | cut -c1-300
echo; cut -d, -f1,5 $O/wgrad_new_256.csv | paste -sd' '; cut -d, -f1,5 $O/wgrad_new_512.csv | paste -sd' '
tail -2 $O/errnew.txt
