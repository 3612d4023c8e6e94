#!/bin/bash
set -u
mkdir -p gpurun_out/c9
timeout 500 python tools/distill_bench.py --steps 4 --native-variants flash+tn+graph,flash+tn > gpurun_out/c9/distill.txt 2> gpurun_out/c9/distill.err; grep '^{' gpurun_out/c9/distill.txt | cut -c1-330; grep -i "warn\|fail" gpurun_out/c9/distill.err | head -5
