#!/bin/bash
# round 5, call 15: the default bench line of the library with the direct GroupNorm statistics form, with the partial-sums form timed on the
# same box before and after it (UNet step only) and the VAE decode both ways.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c15
mkdir -p $O
cd $R
ab() { T2V_GN_CS_DIRECT=$1 timeout 300 python bench.py --steps 40 --warmup 5 --clip 0 --cpu-baseline 0 --distill 0 --breakdown 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); print('direct=$1 ms_per_step', j['ms_per_step'])"; }
ab 0 | tee $O/step_ab.txt
ab 1 | tee -a $O/step_ab.txt
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
ab 0 | tee -a $O/step_ab.txt
ab 1 | tee -a $O/step_ab.txt
for d in 0 1; do T2V_GN_CS_DIRECT=$d timeout 200 python tools/vae_time.py --parity 0 2>/dev/null | tail -1 | sed "s/^/direct=$d /"; done | tee $O/vae_ab.txt
python3 -c "
import json
j=json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['roofline']['frac'], j['clip_4step']['ms'], j['clip_4step']['vae_decode_roofline'], j['clip_16step_v2'].get('ms'), j['distill_step'].get('ms_per_step', j['distill_step']))"
