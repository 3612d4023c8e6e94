#!/bin/bash
set -u
mkdir -p gpurun_out/c26
timeout 400 python tools/distill_bench.py --steps 4 --module-route 1 > gpurun_out/c26/module.txt 2> gpurun_out/c26/module.err; grep '^{' gpurun_out/c26/module.txt | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('module', r['ms_per_step'], r['host_ms_last_step'])"
timeout 400 python tools/distill_bench.py --steps 4 --native-student 1 > gpurun_out/c26/explicit.txt 2> gpurun_out/c26/explicit.err; grep '^{' gpurun_out/c26/explicit.txt | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('explicit', r['ms_per_step'], r['host_ms_last_step'])"
