#!/bin/bash
# Round-2 GPU call 1: the training-path kernels that round 1 never ran on hardware, plus fill-rate microbench.
set -u
mkdir -p gpurun_out/c1
hipcc --offload-arch=gfx950 -O3 tools/fill_rate.hip -o /tmp/fill_rate && (timeout 60 /tmp/fill_rate 16 2000; timeout 60 /tmp/fill_rate 16 2000 64) > gpurun_out/c1/fill_rate.txt 2>&1
T2V_TEST_UNVALIDATED=1 timeout 600 python -m pytest tests/test_gpu_unet_grad.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/c1/unet_grad.txt 2>&1
tail -60 gpurun_out/c1/unet_grad.txt
T2V_TEST_EXPERIMENTAL_TILES=1 timeout 240 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=line -p no:cacheprovider \
    -k "linear_tiles or conv_modes or geglu_all" > gpurun_out/c1/experimental_tiles.txt 2>&1
tail -8 gpurun_out/c1/experimental_tiles.txt
cat gpurun_out/c1/fill_rate.txt | tail -30
