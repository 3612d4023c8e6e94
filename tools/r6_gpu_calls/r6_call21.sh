#!/bin/bash
# round 6, call 21: the v2 16-step clip after the motion-guidance embedding left the sampling loop (one host-to-device copy per step = one
# host / stream synchronisation per step); pipeline fixture test on the device
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c21
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python bench.py --cpu-baseline 0 --distill 0 --breakdown 0 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python3 -c "
import json
j=json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['clip_4step']['ms'], j['clip_4step']['ms_all'], j['clip_16step_v2']['ms'], j['clip_16step_v2']['ms_all'])"
( timeout 900 python -m pytest -q -m gpu tests/test_gpu_engine.py -k "pipeline" 2>&1 | tail -3 ) > $O/t_pipe.txt 2>&1; tail -2 $O/t_pipe.txt
