#!/bin/bash
# The measurements round 1 ran out of GPU budget for, in the order they decide things (≈ 2 GPU-minutes):
#   1. L2 -> CU fill rate: LDS-DMA vs register-staged loads (is 16 B/clk/CU the delivery limit?)
#   2. parity of the compile-verified tile ids 24-29 (the per-tile GEMM tests with the id range extended)
#   3. those ids against the tuned choice on the 30 heaviest UNet shapes, in-graph
# usage: gpurun --timeout 400 -- bash tools/round2_first_call.sh
set -u
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/fill_rate.hip -o /tmp/fill_rate && (timeout 60 /tmp/fill_rate 16 2000; timeout 60 /tmp/fill_rate 16 2000 64) | tee gpurun_out/fill_rate.txt
T2V_TEST_EXPERIMENTAL_TILES=1 timeout 240 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=line -p no:cacheprovider \
    -k "linear_tiles or conv_modes or geglu_all" 2>&1 | tail -15 | tee gpurun_out/experimental_tiles.txt
# 2b. the UNet data-gradient kernels (LayerNorm / GEGLU / temporal-attention backward, two-part GroupNorm backward ...), the
#     gradient engine end to end against autograd, and the LoRA training path (gather kernel, weight-gradient GEMM shapes,
#     student forward + backward with all LoRA gradients)
T2V_TEST_UNVALIDATED=1 timeout 300 python -m pytest tests/test_gpu_unet_grad.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -15 \
    | tee gpurun_out/unet_grad_kernels.txt
timeout 200 python tools/gemm_profile_graph.py --blas 0 --force-cfgs 24,25,26,27,28,29 --top 30 \
    --out gpurun_out/gemm_experimental_cfgs.csv 2>&1 | tail -3
# 4. the distillation step with the native student (tools/distill_bench.py --native-student 1) next to the torch student
T2V_UNVALIDATED_KERNELS=1 timeout 500 python tools/distill_bench.py --steps 3 --native-student 1 2>&1 | tail -2 | tee gpurun_out/distill_native.txt
# 4b. same with each launch list captured in a hipGraph, and a per-kernel profile of the eager run (copy the summary to profiles/)
T2V_UNVALIDATED_KERNELS=1 T2V_HIP_GRAPH=1 timeout 500 python tools/distill_bench.py --steps 3 --native-student 1 2>&1 | tail -1 | tee gpurun_out/distill_native_graph.txt
export TMPDIR=/tmp
T2V_UNVALIDATED_KERNELS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_distill_native -- \
    python tools/distill_bench.py --steps 2 --warmup 1 --native-student 1 > gpurun_out/prof_distill_native.log 2>&1
# (after the above is green) tune the student step's GEMM shapes next to the inference ones:
#   T2V_UNVALIDATED_KERNELS=1 python tools/tune_gemm.py --train 1      (split-K candidates up to 64 for the token-contracted shapes)
# 5. the same step with the flash-style spatial-attention backward (csrc/attention_bwd.hip) instead of the GEMM-formulated one
T2V_UNVALIDATED_KERNELS=1 T2V_FLASH_ATTN_BWD=1 timeout 500 python tools/distill_bench.py --steps 3 --native-student 1 2>&1 | tail -1 | tee gpurun_out/distill_native_flash.txt
# 6. ... and with the token-contracted weight-gradient kernel as well (csrc/wgrad_tn.hip: no operand transposes)
T2V_UNVALIDATED_KERNELS=1 T2V_FLASH_ATTN_BWD=1 T2V_TN_WGRAD=1 timeout 500 python tools/distill_bench.py --steps 3 --native-student 1 2>&1 | tail -1 | tee gpurun_out/distill_native_flash_tn.txt
