#!/bin/bash
# round 3, call 3: where does the LayerNorm fold lose?  per-launch A/B at the UNet's shapes + the fused kernel tests after the fix
set -u
mkdir -p gpurun_out/r3c3
timeout 600 python -m pytest tests/test_gpu_gemm_fuse.py -m gpu -q > gpurun_out/r3c3/fuse_tests.txt 2>&1; tail -4 gpurun_out/r3c3/fuse_tests.txt | cut -c1-300
timeout 600 python tools/fuse_ab.py > gpurun_out/r3c3/fuse_ab.csv 2> gpurun_out/r3c3/fuse_ab.err; cat gpurun_out/r3c3/fuse_ab.csv; tail -3 gpurun_out/r3c3/fuse_ab.err
