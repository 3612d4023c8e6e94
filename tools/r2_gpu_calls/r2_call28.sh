#!/bin/bash
set -u
mkdir -p gpurun_out/c28
timeout 500 python tools/distill_bench.py --steps 5 --native-variants flash+tn,flash+tn+graph > gpurun_out/c28/v.txt 2> gpurun_out/c28/v.err; grep '^{' gpurun_out/c28/v.txt | python -c "
import sys,json
for l in sys.stdin: r=json.loads(l); print(r['variant'], r['ms_per_step'], r['host_ms_last_step'])"
T2V_HIP_GRAPH=1 timeout 500 python tools/distill_bench.py --steps 5 --native-student 1 > gpurun_out/c28/g.txt 2> gpurun_out/c28/g.err; grep '^{' gpurun_out/c28/g.txt | python -c "
import sys,json
for l in sys.stdin: r=json.loads(l); print('T2V_HIP_GRAPH=1 (teacher + student)', r['ms_per_step'], r['host_ms_last_step'])"
timeout 600 python bench.py --cpu-baseline 0 > gpurun_out/c28/bench.json 2> gpurun_out/c28/bench.err; python -c "
import json; r=json.loads(open('gpurun_out/c28/bench.json').read().strip().splitlines()[-1]); print(r['ms_per_step'], r['roofline']['frac'], r['clip_4step']['ms'], r['distill_step']['ms_per_step'])"
