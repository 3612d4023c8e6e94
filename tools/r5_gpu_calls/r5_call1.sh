#!/bin/bash
# round 5, call 1: counters before code (VERDICT r4 "next" 1) + first contact of the round's cheap changes.
#  a. device tests of what changed (row-group LayerNorm, GroupNorm apply with hoisted loads, VAE convs on the halo kernel, dropout scale)
#     and the new parity-chain tests (full-width train mode with replayed masks, C1 geometry on heuristic tiles, mid-width distribution)
#  b. PMC passes (SQ x2, TCC) on ff1 / out / qkv at the 320-channel level
#  c. r05 per-shape in-graph profile of the inference step (GEMM shapes + every other launch class)
#  d. same-box A/B of the whole UNet step: round-4 library vs this one; VAE decode with / without the halo kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c1
mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "layernorm or group_norm or groupnorm or gn_ or halo or dropout" 2>&1 | tail -5 ) > $O/t_kernels.txt 2>&1
( timeout 1500 python -m pytest -q -x -m gpu -s \
    "tests/test_gpu_engine.py::test_unet_full_width_c2_config_vs_oracle" \
    "tests/test_gpu_engine.py::test_unet_full_width_c1_geometry_on_heuristic_tiles" \
    "tests/test_gpu_engine.py::test_vae_decode_full_size_vs_oracle" \
    "tests/test_gpu_train_parity.py::test_mid_width_student_on_device_vs_the_reference_lora_gradient_fixture" \
    "tests/test_gpu_train_parity.py::test_train_mode_student_on_device_with_replayed_masks" \
    "tests/test_gpu_train_parity.py::test_full_width_student_in_train_mode_with_replayed_masks" 2>&1 | grep -v "^$" | tail -40 ) > $O/t_parity.txt 2>&1
tail -3 $O/t_kernels.txt; tail -25 $O/t_parity.txt | cut -c1-400
# d. A/B
for lib in r4 r5 r4 r5; do
  L=$R/t2v-turbo_amd/libt2v_hip.so; [ $lib = r4 ] && L=$R/t2v-turbo_amd/libt2v_hip_r4.so
  T2V_HIP_LIB=$L timeout 300 python bench.py --clip 0 --cpu-baseline 0 --distill 0 --breakdown 0 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'lib':'$lib','ms_per_step':r['ms_per_step'],'parity':r.get('parity')}))"
done | tee $O/step_ab.jsonl
for v in 0 1 0 1; do T2V_VAE_HALO=$v timeout 300 python tools/vae_time.py 2>$O/vae_$v.err | tail -1; done | tee $O/vae_ab.jsonl
# c. profiles
timeout 600 python tools/gemm_profile_graph.py --blas 0 --out $O/gemm_shapes_ingraph.csv > $O/gemm_profile.log 2>&1; tail -2 $O/gemm_profile.log
timeout 600 python tools/op_profile_graph.py --out $O/ops_ingraph.csv > $O/op_profile.log 2>&1; tail -12 $O/op_profile.log
# b. PMC
timeout 120 tools/gemm_lab tools/r5_gpu_calls/spec_timing.txt > $O/lab_timing.csv 2> $O/lab_timing.err; cat $O/lab_timing.csv
cd /tmp
for f in ff1 out qkv; do
  i=0
  for set in "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
             "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
             "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
    i=$((i+1))
    rm -rf /tmp/p_${f}_$i
    timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/p_${f}_$i -- $R/tools/gemm_lab $R/tools/r5_gpu_calls/spec_pmc_$f.txt > /dev/null 2>$O/pmc_${f}_$i.err
    python3 - "$f" "$i" /tmp/p_${f}_$i/*/*counter_collection.csv >> $O/pmc.csv <<'PY'
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
