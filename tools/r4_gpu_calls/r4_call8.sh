#!/bin/bash
# round 4, call 8: bench line (UNet step + clips + distill with the train-mode teacher, C-side replay), then the LoRA epilogue A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c8
mkdir -p $O
cd $R
timeout 1500 python bench.py --cpu-baseline 0 2>$O/bench_a.err | tail -1 > $O/bench_a.json
T2V_LORA_EPILOGUE=1 timeout 900 python bench.py --clip 0 --cpu-baseline 0 --breakdown 0 2>$O/bench_lora.err | tail -1 > $O/bench_lora.json
T2V_C_REPLAY=0 timeout 900 python bench.py --clip 0 --cpu-baseline 0 --breakdown 0 2>$O/bench_pyreplay.err | tail -1 > $O/bench_pyreplay.json
for f in bench_a bench_lora bench_pyreplay; do python - $O/$f.json $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read())
ds=d.get("distill_step",{})
print(sys.argv[2], "unet ms", d.get("ms_per_step"), "| distill", {k:ds.get(k) for k in ("ms_per_step","issue","ms_per_step_by_issue","ms_per_step_eval_teacher","host_ms_last_step")}, "parity", ds.get("parity",{}).get("ok") if isinstance(ds.get("parity"),dict) else None, "| clip4", d.get("clip_4step",{}).get("ms") if isinstance(d.get("clip_4step"),dict) else None)
PY
done
