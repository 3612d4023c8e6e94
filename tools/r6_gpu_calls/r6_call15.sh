#!/bin/bash
# round 6, call 15: where the VAE decode's 25.8 ms sit now: rocprofv3 kernel summary of tools/vae_time.py (no parity leg), per launch in order
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c15
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_vae
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vae -- python $R/tools/vae_time.py --reps 6 --parity 0 > $O/vae_time.log 2>&1
cp $(find /tmp/prof_vae -name "*kernel_stats.csv" | head -1) $O/vae_kernel_stats.csv
python3 - <<PY
import csv, glob
f = glob.glob('/tmp/prof_vae/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last decode: the final 1/8 of the launches (8 decodes: 2 warm + 6 timed)
n = len(rows) // 8
last = rows[-n:]
t0 = int(last[0]['Start_Timestamp'])
with open('$O/vae_last_decode_trace.csv', 'w') as o:
    o.write('i,kernel,start_us,dur_us,gap_before_us\n')
    prev_end = None
    for i, r in enumerate(last):
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = 0 if prev_end is None else (s - prev_end) / 1e3
        o.write(f"{i},{r['Kernel_Name'][:60].replace(',', ';')},{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.1f},{gap:.1f}\n")
        prev_end = e
print('launches per decode', n, 'span ms', (int(last[-1]['End_Timestamp']) - t0) / 1e6)
PY
tail -2 $O/vae_time.log
head -14 $O/vae_kernel_stats.csv | cut -c1-200
