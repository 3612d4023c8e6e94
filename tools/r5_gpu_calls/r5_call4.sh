#!/bin/bash
# round 5, call 4: the halo kernel over a nearest-x2 upsampled source (T2V_GEMM_CONV3X3_UP2): device tests, then same-box A/B of the
# VAE decode and of the UNet step with / without it
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c4
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "halo" > $O/t_kernels.txt 2>&1; tail -3 $O/t_kernels.txt
for v in 0 1 0 1; do T2V_HALO_UP2=$v timeout 300 python tools/vae_time.py --parity $v 2>$O/vae_$v.err | tail -1 | cut -c1-500; done | tee $O/vae_up2_ab.jsonl
for v in 0 1 0 1; do
  T2V_HALO_UP2=$v timeout 300 python bench.py --clip 0 --cpu-baseline 0 --distill 0 --breakdown 0 2>$O/bench_$v.err | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'T2V_HALO_UP2':$v,'ms_per_step':r['ms_per_step'],'parity':r.get('parity')}))"
done | tee $O/step_up2_ab.jsonl
timeout 900 python -m pytest -q -m gpu "tests/test_gpu_engine.py::test_unet_full_width_c2_config_vs_oracle" "tests/test_gpu_engine.py::test_vae_decode_full_size_vs_oracle" > $O/t_engine.txt 2>&1; tail -3 $O/t_engine.txt
