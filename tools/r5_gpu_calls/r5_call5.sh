#!/bin/bash
# round 5, call 5: the direct small-Cout conv (VAE conv_out) against the implicit-GEMM conv it replaces, same box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c5
mkdir -p $O
cd $R
for v in 0 1 0 1; do T2V_SMALL_COUT=$v timeout 300 python tools/vae_time.py --parity $v 2>$O/vae_$v.err | tail -1 | cut -c1-500; done | tee $O/vae_small_cout_ab.jsonl
