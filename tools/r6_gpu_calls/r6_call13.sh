#!/bin/bash
# round 6, call 13: state of the tree after ln_in + the T2V_EXPERIMENTAL split + the in-graph roofline of the bench:
# full device suite, the default bench line, per-shape in-graph GEMM table, per-op in-graph table, rocprofv3 kernel summary of the bench loop
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c13
mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 ) > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python3 -c "
import json
j=json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['roofline'], j.get('clip_4step'), {k: v for k, v in j.get('distill_step', {}).items() if k in ('ms_per_step','parity')})"
timeout 600 python tools/gemm_profile_graph.py --blas 0 --out $O/gemm_shapes_ingraph.csv > $O/gemm_shapes.log 2>&1; tail -1 $O/gemm_shapes.log
timeout 600 python tools/op_profile_graph.py --out $O/ops_ingraph.csv > $O/ops.log 2>&1; tail -2 $O/ops.log
cd /tmp
rm -rf /tmp/prof_stats
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --steps 10 --warmup 2 --clip 0 --cpu-baseline 0 --distill 0 --graph 0 --breakdown 0 > $O/prof_stats.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/unet_bench_kernel_stats.csv
head -12 $O/unet_bench_kernel_stats.csv
