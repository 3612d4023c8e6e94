#!/bin/bash
# round 4, call 13: where does the LoRA epilogue's time go? (ablation build, torch-free lab)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${RUN:-r4c13}
mkdir -p $O
cd $R
export T2V_LAB_LIBS=$R/t2v-turbo_amd/libt2v_hip.so:$R/t2v-turbo_amd/libt2v_hip_ablate.so
timeout 200 tools/gemm_lab tools/r4_gpu_calls/spec_lora_epi.txt > $O/lora_epi.csv 2> $O/lora_epi.err
cat $O/lora_epi.csv; tail -3 $O/lora_epi.err
