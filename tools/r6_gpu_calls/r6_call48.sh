#!/bin/bash
# round 6, call 48: the device test of full fine-tuning with three weight updates (eager refresh, captured refresh, replayed refresh) + the bench leg alone
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c48
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_train_parity.py -q -x -m gpu -s -k "full_fine_tuning_on_device or full_fine_tuning_mid or full_fine_tuning_train" 2>&1 | grep -v "^$" | tail -12 | tee $O/pytest_full.txt
timeout 600 python bench.py --steps 3 --warmup 1 --clip 0 --cpu-baseline 0 --distill 0 --full-finetune 1 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; grep "full fine-tuning leg" $O/bench.err | cut -c1-500
