#!/bin/bash
set -u
mkdir -p gpurun_out/c8
timeout 300 python -m pytest tests/test_gpu_unet_grad.py -m gpu -q --tb=short -p no:cacheprovider -k "wgrad_tn or lora_training_engine" > gpurun_out/c8/wgrad.txt 2>&1; tail -4 gpurun_out/c8/wgrad.txt | cut -c1-300
timeout 400 python tools/distill_bench.py --steps 3 --native-variants flash+tn > gpurun_out/c8/distill.txt 2> gpurun_out/c8/distill.err; grep '^{' gpurun_out/c8/distill.txt
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c8 -- python $GRAFT_REPO_ROOT/tools/distill_bench.py --steps 2 --warmup 2 --native-variants flash+tn > $GRAFT_REPO_ROOT/gpurun_out/c8/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_window.py /tmp/prof_c8 --marker sinh --steps 2 --out $GRAFT_REPO_ROOT/gpurun_out/c8/distill_window_stats.csv
head -45 $GRAFT_REPO_ROOT/gpurun_out/c8/distill_window_stats.csv | cut -c1-200
