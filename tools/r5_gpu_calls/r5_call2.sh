#!/bin/bash
# round 5, call 2: what call 1 did not get to (a pytest -x stopped at a test-side assertion; gemm_lab was started from /tmp without the
# library path) + per-class A/B of the round-4 and round-5 libraries (the whole step came out 0.17 ms SLOWER with the new norm kernels)
# + the 64-channel wave-tile variant of the halo kernel on the VAE decoder
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c2
mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "halo" 2>&1 | tail -5 ) > $O/t_kernels.txt 2>&1
tail -3 $O/t_kernels.txt
for lib in r4 r5; do
  L=$R/t2v-turbo_amd/libt2v_hip.so; [ $lib = r4 ] && L=$R/t2v-turbo_amd/libt2v_hip_r4.so
  T2V_HIP_LIB=$L timeout 400 python tools/op_profile_graph.py --out $O/ops_ingraph_$lib.csv > $O/op_profile_$lib.log 2>&1; head -8 $O/ops_ingraph_$lib.csv | cut -c1-120
done
for v in 0 1 0 1; do T2V_VAE_HALO=$v timeout 300 python tools/vae_time.py --parity $v 2>$O/vae_$v.err | tail -1 | cut -c1-400; done | tee $O/vae_ab.jsonl
for lib in r4 r5 r4 r5; do
  L=$R/t2v-turbo_amd/libt2v_hip.so; [ $lib = r4 ] && L=$R/t2v-turbo_amd/libt2v_hip_r4.so
  T2V_HIP_LIB=$L timeout 300 python bench.py --clip 0 --cpu-baseline 0 --distill 0 --breakdown 0 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'lib':'$lib','ms_per_step':r['ms_per_step']}))"
done | tee $O/step_ab.jsonl
( timeout 2400 python -m pytest -q -m gpu -s \
    "tests/test_gpu_engine.py::test_unet_full_width_c1_geometry_off_the_tuned_table" \
    "tests/test_gpu_engine.py::test_vae_decode_full_size_vs_oracle" \
    "tests/test_gpu_train_parity.py::test_mid_width_student_on_device_vs_the_reference_lora_gradient_fixture" \
    "tests/test_gpu_train_parity.py::test_train_mode_student_on_device_with_replayed_masks" \
    "tests/test_gpu_train_parity.py::test_full_width_student_in_train_mode_with_replayed_masks" 2>&1 | grep -v "^$" | tail -60 ) > $O/t_parity.txt 2>&1
grep -n "C1 geometry\|mid-width\|train mode\|passed\|failed\|Error\|error\|full width" $O/t_parity.txt | cut -c1-600 | tail -30
export T2V_LAB_LIBS=$R/t2v-turbo_amd/libt2v_hip.so
cd /tmp
for f in ff1 out qkv; do
  i=0
  for set in "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
             "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
             "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
    i=$((i+1))
    rm -rf /tmp/p_${f}_$i
    timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/p_${f}_$i -- $R/tools/gemm_lab $R/tools/r5_gpu_calls/spec_pmc_$f.txt > /dev/null 2>$O/pmc_${f}_$i.err
    python3 - "$f" "$i" $(find /tmp/p_${f}_$i -name "*counter_collection.csv" | head -1) >> $O/pmc.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[3])):
    if "gemm_kernel" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"{sys.argv[1]},pass{sys.argv[2]},{k},{acc[k][0] / acc[k][1]:.0f},{acc[k][1]}")
PY
  done
done
cat $O/pmc.csv
