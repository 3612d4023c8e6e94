#!/bin/bash
# Variant libraries that differ from the product library in attention.hip's compile flags only (tools/attn_variants_time.sh times them):
# the other objects are the product build's.  Usage: bash tools/build_attn_variants.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/t2v-turbo_amd/csrc
OBJS=$(ls $C/*.o | grep -v "variant\|attention.o\|attention\.v")
i=0
# (variant list: one flag set per line in $ATTN_VARIANTS, default = the round-5 issue-order set)
DEFAULT_VARIANTS=$'\n-DT2V_ATTN_SETPRIO\n-DT2V_ATTN_NOFENCE\n-DT2V_ATTN_SETPRIO -DT2V_ATTN_NOFENCE\n-DT2V_ATTN_WPE=2\n-DT2V_ATTN_WPE=2 -DT2V_ATTN_SETPRIO'
while IFS= read -r flags; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I $R/include -I $C $flags -c $C/attention.hip -o $C/attention.v$i.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/t2v-turbo_amd/libt2v_hip_attn$i.so $OBJS $C/attention.v$i.o
  echo "variant $i: $flags"
  i=$((i+1))
done <<< "${ATTN_VARIANTS-$DEFAULT_VARIANTS}"
