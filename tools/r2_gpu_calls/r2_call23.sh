#!/bin/bash
set -u
mkdir -p gpurun_out/c23
timeout 400 python -m pytest tests/test_gpu_engine.py -m gpu -q --tb=short -p no:cacheprovider -k "full_width" > gpurun_out/c23/fw.txt 2>&1; tail -2 gpurun_out/c23/fw.txt | cut -c1-200
for i in 1 2; do timeout 400 python tools/distill_bench.py --steps 6 --native-student 1 > gpurun_out/c23/d$i.txt 2> gpurun_out/c23/d$i.err; grep '^{' gpurun_out/c23/d$i.txt | cut -c150-330; done
timeout 600 python bench.py --clip 0 --cpu-baseline 0 > gpurun_out/c23/bench.json 2> gpurun_out/c23/bench.err; python -c "
import json; r=json.loads(open('gpurun_out/c23/bench.json').read().strip().splitlines()[-1]); print(r['ms_per_step'], r['distill_step']['ms_per_step'], r['distill_step']['forward_ms'], r['distill_step']['backward_ms'])"
