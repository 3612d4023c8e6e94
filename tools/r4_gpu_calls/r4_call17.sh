#!/bin/bash
# (variant libraries: csrc/wgrad_tn.hip of that experiment compiled with -DT2V_WGRAD_AHEAD=2 / 3 and linked with the product objects; neither the patch nor the libraries are in the tree)
# round 4, call 17: t2v_wgrad_tn_group (64 x 64 tiles) with one / two / three token steps of loads in flight, and with more blocks
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c17
mkdir -p $O
cd $R
P=$R/t2v-turbo_amd
timeout 250 python tools/wgrad_time.py $P/libt2v_hip.so $P/libt2v_hip_wgahead2.so $P/libt2v_hip_wgahead3.so > $O/ahead.csv 2> $O/err.txt
cut -d, -f1,5,7,9,11 $O/ahead.csv
for B in 960 1280; do T2V_WGRAD_BLOCKS=$B timeout 250 python tools/wgrad_time.py $P/libt2v_hip.so $P/libt2v_hip_wgahead2.so > $O/ahead_b$B.csv 2>> $O/err.txt; echo "blocks $B"; cut -d, -f1,5,7 $O/ahead_b$B.csv | paste -sd' '; done
tail -2 $O/err.txt
