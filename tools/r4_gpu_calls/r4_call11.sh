#!/bin/bash
# round 4, call 11: 16-bit-per-element dropout mask + split-aware / halo-aware choice of the LoRA form: device tests, per-shape student
# GEMM profile, distillation step A/B (same box)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c11
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_gemm_fuse.py tests/test_gpu_kernels.py tests/test_gpu_train_parity.py -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout 300 python tools/student_gemm_profile.py > $O/student_gemm_shapes.csv 2> $O/prof.err
tail -1 $O/student_gemm_shapes.csv
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --clip 0 --cpu-baseline 0 --breakdown 0 2>$O/$name.err | tail -1 > $O/$name.json
  python - $O/$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read())
ds=d.get("distill_step",{})
print(sys.argv[2], "unet ms", d.get("ms_per_step"), "| distill", {k:ds.get(k) for k in ("ms_per_step","ms_per_step_by_issue","forward_ms","backward_ms","launches")})
PY
}
run new_default T2V_X=0
run old_choice T2V_LORA_SPLIT_AWARE=0 T2V_LORA_HALO_MIN_C=0
run halo320 T2V_LORA_HALO_MIN_C=320
run split_only T2V_LORA_HALO_MIN_C=0
