#!/bin/bash
# round 4, call 12: LoRA epilogue with the U tile in LDS: kernel tests, per-shape profile, distillation step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c12
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_gemm_fuse.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
timeout 300 python tools/student_gemm_profile.py > $O/student_gemm_shapes.csv 2> $O/prof.err
tail -1 $O/student_gemm_shapes.csv
env T2V_X=0 timeout 600 python bench.py --clip 0 --cpu-baseline 0 --breakdown 0 2>$O/bench.err | tail -1 > $O/bench.json
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read())
ds=d.get("distill_step",{})
print("unet ms", d.get("ms_per_step"), "| distill", {k:ds.get(k) for k in ("ms_per_step","ms_per_step_by_issue","forward_ms","backward_ms","launches")})
PY
