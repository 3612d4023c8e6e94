#!/bin/bash
# round 5, call 18 (last GPU seconds of the round): the UNet leg of the bench on the final library and the rocprofv3 kernel summary of the
# same loop on the same box (the full default line needs 3.5 minutes: not left).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c18
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 120 python bench.py --clip 0 --cpu-baseline 0 --distill 0 > $O/bench_line_unet_leg.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp
rm -rf /tmp/prof_stats
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --steps 10 --warmup 2 --clip 0 --cpu-baseline 0 --distill 0 --graph 0 > $O/prof_stats.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/unet_bench_kernel_stats.csv
python3 -c "
import json
j=json.loads(open('$O/bench_line_unet_leg.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['roofline']['frac'], {k: (v['launches'], v['ms']) for k, v in j['kernel_ms'].items() if 'norm' in k})"
