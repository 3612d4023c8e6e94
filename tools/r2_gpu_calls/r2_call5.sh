#!/bin/bash
set -u
mkdir -p gpurun_out/c5
timeout 400 python tools/distill_nan_probe.py --steps 6 --phases C > gpurun_out/c5/nan_probe.txt 2> gpurun_out/c5/nan_probe.err
grep -E "^phase|non-finite|student out|parameters" gpurun_out/c5/nan_probe.txt | head -20; tail -2 gpurun_out/c5/nan_probe.err
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_unet_grad.py -m gpu -q -x --tb=short -p no:cacheprovider -s > gpurun_out/c5/tests_engine.txt 2>&1; tail -15 gpurun_out/c5/tests_engine.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "mean_far or fill_zero or groupnorm or group_norm or layernorm" > gpurun_out/c5/tests_kernels.txt 2>&1; tail -15 gpurun_out/c5/tests_kernels.txt
timeout 400 python tools/distill_bench.py --steps 3 --module-route 1 > gpurun_out/c5/module_route.txt 2> gpurun_out/c5/module_route.err; tail -1 gpurun_out/c5/module_route.txt; tail -4 gpurun_out/c5/module_route.err
