#!/bin/bash
# A variant of libt2v_hip.so that differs from the product in csrc/linear_pr.hip's build-time knobs only: that one file is compiled
# with the given flags and linked against the product's other objects (seconds instead of the two minutes of a full variant build).
#   tools/build_lpr_variant.sh <tag> [hipcc flags...]   ->  t2v-turbo_amd/libt2v_hip_<tag>.so
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/t2v-turbo_amd/csrc
tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I $R/include -I $C "$@" -c $C/linear_pr.hip -o $C/linear_pr.$tag.variant.o
objs=$(ls $C/*.o | grep -v "\.variant\.o$" | grep -v "/linear_pr\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/t2v-turbo_amd/libt2v_hip_$tag.so $objs $C/linear_pr.$tag.variant.o
echo $R/t2v-turbo_amd/libt2v_hip_$tag.so
