#!/bin/bash
# round 6, call 30: the full fine-tuning step after the final-reduction kernel of t2v_norm_affine_grad was rebuilt (one thread per value walking ~1 000
# partial blocks -> 16 waves per 64 values): kernel tests, the full-width parity test, the step time
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c30
mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "affine_grad or im2col" 2>&1 | tail -3 ) > $O/t_kernels.txt 2>&1; tail -2 $O/t_kernels.txt
( timeout 1500 python -m pytest tests/test_gpu_train_parity.py -q -m gpu -s -k "full_fine_tuning" 2>&1 | grep -v "^$" | tail -10 ) > $O/t_full.txt 2>&1; tail -4 $O/t_full.txt | cut -c1-300
timeout 900 python tools/full_finetune_time.py --frames 16 --steps 4 > $O/full_finetune.json 2> $O/ff.err; tail -1 $O/full_finetune.json | cut -c1-500
