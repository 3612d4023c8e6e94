#!/bin/bash
# round 4, call 23 (last GPU seconds): flash attention forward / backward with the XCD-contiguous block order: device tests, UNet step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c23
mkdir -p $O
cd $R
timeout 55 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet_grad.py -x -q -m gpu -k "attn or attention" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
timeout 100 python bench.py --steps 20 --clip 0 --distill 0 --cpu-baseline 0 2>$O/err.txt | tail -1 > $O/bench_unet.json
python - $O/bench_unet.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); k=d["kernel_ms"]
print("unet ms", d["ms_per_step"], "parity", d.get("parity_rel_l2"), {n:k[n] for n in k if "attn" in n})
PY
