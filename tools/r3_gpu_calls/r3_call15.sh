#!/bin/bash
# round 3, call 15: dropout epilogue on the fast kernels (FUSE bit 8): tests incl. the full-width training parity gate, per-shape
# GEMM profile of the student with it on / off, distillation step on / off / with hipGraph replay of the student's lists
set -u
O=gpurun_out/r3c15
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_unet_grad.py tests/test_gpu_gemm_fuse.py tests/test_gpu_train_parity.py -m gpu -q -x > $O/tests.txt 2>&1
grep -E "passed|failed" $O/tests.txt | tail -2
T2V_GEMM_FAST_DROPOUT=1 timeout 600 python tools/student_gemm_profile.py > $O/student_gemm_fastdrop1.csv 2> $O/p1.err
T2V_GEMM_FAST_DROPOUT=0 timeout 600 python tools/student_gemm_profile.py > $O/student_gemm_fastdrop0.csv 2> $O/p0.err
tail -1 $O/student_gemm_fastdrop1.csv; tail -1 $O/student_gemm_fastdrop0.csv
python - <<'PY'
import csv
def load(f):
    d = {}
    for r in csv.reader(open(f)):
        if r and r[0] in ('rec', 'rec_bwd') and r[7] == '1':
            d[tuple(r[:11])] = (int(r[11]), float(r[12]))
    return d
a, b = load('gpurun_out/r3c15/student_gemm_fastdrop1.csv'), load('gpurun_out/r3c15/student_gemm_fastdrop0.csv')
ta = tb = 0
for k in sorted(a, key=lambda k: -b.get(k, (0, 0))[0] * b.get(k, (0, 0))[1]):
    if k in b:
        ta += a[k][0] * a[k][1]; tb += b[k][0] * b[k][1]
        print(','.join(k), 'count', a[k][0], 'fast', a[k][1], 'generic', b[k][1])
print('dropout-epilogue launches per forward: fast %.2f ms, generic %.2f ms' % (ta / 1e3, tb / 1e3))
PY
for v in 1 0 1 0; do
  T2V_GEMM_FAST_DROPOUT=$v timeout 600 python tools/distill_bench.py --native-student 1 --steps 6 --warmup 2 2> $O/d$v.err | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('fastdrop=$v', d['ms_per_step'], d['host_ms_last_step'])
"
done
T2V_GEMM_FAST_DROPOUT=1 timeout 600 python tools/distill_bench.py --native-student 1 --steps 6 --warmup 2 --native-variants "flash+tn+graph" 2> $O/dg.err | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('fastdrop=1+graph', d['ms_per_step'], d['host_ms_last_step'])
"
