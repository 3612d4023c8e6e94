#!/bin/bash
# round 5, call 3: re-run what failed in call 2 for test-side reasons (full output kept this time), where the VAE decode's time goes
# (rocprofv3 kernel stats), same-box step A/B of the library with the row-group LayerNorm only, and the nearest-shape tile table on the
# distillation step (its LoRA / backward shapes are only partly in the table)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c3
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "halo or layernorm" > $O/t_kernels.txt 2>&1; tail -3 $O/t_kernels.txt
timeout 1500 python -m pytest -q -m gpu -s \
    "tests/test_gpu_engine.py::test_unet_full_width_c1_geometry_off_the_tuned_table" \
    "tests/test_gpu_train_parity.py::test_mid_width_student_on_device_vs_the_reference_lora_gradient_fixture" > $O/t_parity.txt 2>&1
grep -n "C1 geometry\|tile-table\|mid-width\|passed\|failed\|Error" $O/t_parity.txt | cut -c1-900
for lib in r4 r5 r4 r5; do
  L=$R/t2v-turbo_amd/libt2v_hip.so; [ $lib = r4 ] && L=$R/t2v-turbo_amd/libt2v_hip_r4.so
  T2V_HIP_LIB=$L timeout 300 python bench.py --clip 0 --cpu-baseline 0 --distill 0 --breakdown 0 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'lib':'$lib','ms_per_step':r['ms_per_step']}))"
done | tee $O/step_ab.jsonl
for v in 0 1 0 1; do
  T2V_GEMM_TUNE_NEAREST=$v timeout 400 python bench.py --clip 0 --cpu-baseline 0 --breakdown 0 --distill-parity 0 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); ds=r['distill_step']
print(json.dumps({'T2V_GEMM_TUNE_NEAREST':$v,'unet_ms':r['ms_per_step'],'distill_ms':ds['ms_per_step'],'by_issue':ds.get('ms_per_step_by_issue'),'forward_ms':ds.get('forward_ms'),'backward_ms':ds.get('backward_ms')}))"
done | tee $O/distill_nearest_ab.jsonl
cd /tmp
rm -rf /tmp/prof_vae
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vae -- python $R/tools/vae_time.py --parity 0 --reps 10 > $O/prof_vae.log 2>&1
cp $(find /tmp/prof_vae -name "*kernel_stats.csv" | head -1) $O/vae_decode_kernel_stats.csv
head -25 $O/vae_decode_kernel_stats.csv | cut -c1-200
