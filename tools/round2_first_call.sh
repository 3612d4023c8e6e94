#!/bin/bash
# First GPU call of round 2 (~10 GPU-minutes): everything round 1 wrote after its GPU budget was spent, in the order it decides things.
#   1. L2 -> CU fill rate: LDS-DMA vs register-staged loads (is 16 B/clk/CU the delivery limit?)
#   2. parity of the tile ids 24-29 of t2v_gemm (functionally verified on the host simulator only)
#   3. every kernel of the UNet gradient / LoRA training path (backward_unet.hip, train.hip, attention_bwd.hip, wgrad_tn.hip) against
#      the emulated backend, the gradient engine and the LoRA training engine end to end against autograd
#   4. tile ids 24-29 against the tuned choice on the 30 heaviest UNet shapes, in-graph
# usage: gpurun --timeout 900 -- bash tools/round2_first_call.sh        then: tools/round2_second_call.sh
set -u
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/fill_rate.hip -o /tmp/fill_rate && (timeout 60 /tmp/fill_rate 16 2000; timeout 60 /tmp/fill_rate 16 2000 64) | tee gpurun_out/fill_rate.txt
T2V_TEST_EXPERIMENTAL_TILES=1 timeout 240 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=line -p no:cacheprovider \
    -k "linear_tiles or conv_modes or geglu_all" 2>&1 | tail -15 | tee gpurun_out/experimental_tiles.txt
T2V_TEST_UNVALIDATED=1 timeout 300 python -m pytest tests/test_gpu_unet_grad.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -25 \
    | tee gpurun_out/unet_grad_kernels.txt
timeout 200 python tools/gemm_profile_graph.py --blas 0 --force-cfgs 24,25,26,27,28,29 --top 30 \
    --out gpurun_out/gemm_experimental_cfgs.csv 2>&1 | tail -3
