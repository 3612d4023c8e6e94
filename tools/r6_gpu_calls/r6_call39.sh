#!/bin/bash
# round 6, call 39: t2v_repack_conv_f32 (conv packs re-made per step by a library kernel): kernel tests, step time, kernel summary, parity tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${RUN:-r6c39}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "repack or im2col or affine" 2>&1 | tail -3 | tee $O/pytest_kernels.txt
bash tools/r6_gpu_calls/r6_call37.sh
