#!/bin/bash
set -u
mkdir -p gpurun_out/c29
for bt in 0 1 0 1; do
timeout 400 python tools/distill_bench.py --steps 6 --native-student 1 --batch-teacher $bt > gpurun_out/c29/d.txt 2> gpurun_out/c29/d.err; grep '^{' gpurun_out/c29/d.txt | python -c "
import sys,json
for l in sys.stdin: r=json.loads(l); print('batch_teacher=$bt', r['ms_per_step'], r['host_ms_last_step'])"
done
