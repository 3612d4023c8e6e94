#!/bin/bash
# round 5, call 16: after routing the VAE decoder's 256- / 128-channel norms back to the partial-sums statistics form: the decode both
# ways (T2V_GN_CS_DIRECT=0 = every norm on the partial-sums form), the device tests of the decoder and of the op
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c16
mkdir -p $O
cd $R
for d in 0 1 0 1; do T2V_GN_CS_DIRECT=$d timeout 200 python tools/vae_time.py --parity 0 2>/dev/null | tail -1 | cut -c1-80 | sed "s/^/direct=$d /"; done | tee $O/vae_ab.txt
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -q -m gpu -k "vae or norm" 2>&1 | tail -2 | tee $O/tests.txt
