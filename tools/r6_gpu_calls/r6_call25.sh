#!/bin/bash
# round 6, call 25: the full-width TRAIN-mode parity gate (tests/test_gpu_train_parity.py::test_full_width_student_in_train_mode_with_replayed_masks)
# at MORE frames than the suite's 2 (VERDICT r5 "what's weak" 1(i)): 8 frames, then the timed 16 — as far as the host side (fp32 CPU autograd at
# full width + ~1e9 mask elements) fits the box and the time limit.  Evidence run, not part of the suite.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c25
mkdir -p $O
cd $R
export TMPDIR=/tmp
free -g | head -2 > $O/mem.txt; nproc >> $O/mem.txt; cat $O/mem.txt
for f in 8 16; do
  timeout 1500 python - $f > $O/frames_$f.txt 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
import bench
from t2v_turbo_amd.native import HipOps
from tests.test_gpu_train_parity import run_student_train_mode_vs_reference_oracle, OUT_TOL, DX_TOL
f = int(sys.argv[1])
torch.set_num_threads(64)
t0 = time.time()
run_student_train_mode_vs_reference_oracle(torch.device("cuda", 0), HipOps(), bench.VC2_UNET, (1, 4, f, 40, 64), 1150, 400, OUT_TOL, DX_TOL, 0.985, 0.12, seed_model=4321)
print(f"FRAMES {f} OK in {time.time() - t0:.0f} s")
PY
  echo "frames $f rc=$?"; grep -v "^$" $O/frames_$f.txt | tail -6 | cut -c1-300
done
