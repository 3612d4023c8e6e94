#!/bin/bash
set -u
mkdir -p gpurun_out/c21
timeout 400 python tools/distill_bench.py --steps 4 --module-route 1 > gpurun_out/c21/module.txt 2> gpurun_out/c21/module.err; grep '^{' gpurun_out/c21/module.txt | cut -c1-300; tail -2 gpurun_out/c21/module.err | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/c21/pytest_gpu.txt 2>&1; tail -3 gpurun_out/c21/pytest_gpu.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
