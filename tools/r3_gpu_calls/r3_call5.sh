#!/bin/bash
# round 3, call 5: LayerNorm fold through the accumulators' initial value (epilogue = one multiply) — per-launch A/B again + tests
set -u
mkdir -p gpurun_out/r3c5
T2V_AB_TILES=23,31,11,7,4 timeout 300 python tools/fuse_ab.py 2>/dev/null | grep -E "qkv|cross_q|ff1" | tee gpurun_out/r3c5/ab.csv

