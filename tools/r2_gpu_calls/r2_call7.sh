#!/bin/bash
set -u
mkdir -p gpurun_out/c7
timeout 300 python -m pytest tests/test_gpu_engine.py -m gpu -q --tb=short -p no:cacheprovider -k "ddim_inversion or alternating" > gpurun_out/c7/f4.txt 2>&1; tail -6 gpurun_out/c7/f4.txt | cut -c1-300
T2V_TEST_EXPERIMENTAL_TILES=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=line -p no:cacheprovider -k "linear_tiles or conv_modes or geglu_all" > gpurun_out/c7/tiles.txt 2>&1; tail -5 gpurun_out/c7/tiles.txt | cut -c1-300
timeout 400 python tools/gemm_profile_graph.py --blas 0 --force-cfgs 30,31,32,33 --top 70 --out gpurun_out/c7/gemm_8wave_cfgs.csv > gpurun_out/c7/gemm_exp.log 2>&1
tail -2 gpurun_out/c7/gemm_exp.log | cut -c1-300
