#!/bin/bash
# round 4, call 2: first hardware contact of t2v_conv_halo: sampled fp64 check + time against the tuned t2v_gemm tiles
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c2
mkdir -p $O
cd $R
timeout 300 tools/gemm_lab tools/r4_gpu_calls/spec_halo1.txt > $O/halo1.csv 2> $O/halo1.err
cat $O/halo1.csv
tail -5 $O/halo1.err
