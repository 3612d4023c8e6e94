#!/bin/bash
set -u
mkdir -p gpurun_out/c4
timeout 300 python tools/graph_bisect.py --tiny 1 > gpurun_out/c4/bisect_tiny.txt 2>&1; tail -25 gpurun_out/c4/bisect_tiny.txt
timeout 300 python tools/graph_bisect.py --tiny 1 --which rec_bwd > gpurun_out/c4/bisect_tiny_bwd.txt 2>&1; tail -12 gpurun_out/c4/bisect_tiny_bwd.txt
