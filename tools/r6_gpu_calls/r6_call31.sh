#!/bin/bash
# round 6, call 31: hipGraph capture in thread-local mode (the one-rank RCCL test aborted once in five full-suite runs: the process group's watchdog
# thread polled an event while the engine captured a graph in the default global mode): the RCCL tests five times, every test that captures a
# graph, the UNet leg of the bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c31
mkdir -p $O
cd $R
export TMPDIR=/tmp
for i in 1 2 3 4 5; do
  ( timeout 900 python -m pytest tests -q -m gpu -k "rccl" 2>&1 | tail -2 ) > $O/rccl_$i.txt 2>&1; tail -1 $O/rccl_$i.txt
done
( timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_train_parity.py -q -m gpu -k "graph or c2_config or plan or trainer or overlapped" 2>&1 | tail -4 ) > $O/t_graph.txt 2>&1; tail -2 $O/t_graph.txt
timeout 400 python bench.py --clip 0 --cpu-baseline 0 --distill 0 --full-finetune 0 > $O/bench_unet.json 2> $O/bench.err; python3 -c "
import json
j=json.loads(open('$O/bench_unet.json').read().strip().splitlines()[-1]); print(j['ms_per_step'], j['config'].get('graph_captured'), j['roofline']['frac'], j['roofline']['timing'][:40])"
