#!/bin/bash
# round 6, call 28: the default bench line with the new full_finetune_step leg (total run time, memory after the distillation leg)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c28
mkdir -p $O
cd $R
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"
python3 -c "
import json
j=json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['roofline']['frac'], j['clip_4step']['ms'], j['distill_step']['ms_per_step'], j.get('full_finetune_step'))"
tail -3 $O/bench.err | cut -c1-300
