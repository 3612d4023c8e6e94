#!/usr/bin/env python
"""Inter-kernel gaps of the hipGraph-replayed UNet step from a rocprofv3 --kernel-trace CSV.
    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python bench.py --steps 3 --warmup 1 --clip 0 --cpu-baseline 0
    python tools/gap_analysis.py /tmp/kt/*/*kernel_trace.csv --launches 787"""
import argparse
import csv
import collections

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--launches", type=int, required=True)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.trace)))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
ks = [k for k in ks if "anonymous namespace" in k[2] and "at::" not in k[2]]
n_groups = len(ks) // a.launches
for gi in range(min(n_groups, 8), 0, -1):  # per-step span / busy for the last few steps (oldest first)
    grp = ks[len(ks) - gi * a.launches: len(ks) - (gi - 1) * a.launches]
    print(f"step -{gi}: span {(grp[-1][1] - grp[0][0]) / 1e6:.3f} ms, busy {sum(e - s for s, e, _ in grp) / 1e6:.3f} ms, "
          f"idle before {(grp[0][0] - ks[len(ks) - gi * a.launches - 1][1]) / 1e6 if len(ks) > gi * a.launches else 0:.3f} ms")
last = ks[-a.launches:]
busy = sum(e - s for s, e, _ in last)
span = last[-1][1] - last[0][0]
gaps = [(last[i][0] - last[i - 1][1], last[i - 1][2], last[i][2]) for i in range(1, len(last))]
print(f"last step: span {span / 1e6:.3f} ms, kernel busy {busy / 1e6:.3f} ms, gaps {sum(g for g, _, _ in gaps) / 1e6:.3f} ms over {len(gaps)} boundaries")
hist = collections.Counter(min(int(g / 1000), 10) for g, _, _ in gaps)
print("gap histogram (us bucket: count):", sorted(hist.items()))
by = collections.defaultdict(lambda: [0, 0])
for g, prev, nxt in gaps:
    key = nxt.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
    by[key][0] += 1
    by[key][1] += g
for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  before {k:50s} n={n:4d} total {t / 1e6:.3f} ms avg {t / n / 1e3:.2f} us")
dur = collections.defaultdict(lambda: [0, 0])
for s, e, n in last:
    key = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
    dur[key][0] += 1
    dur[key][1] += e - s
for k, (n, t) in sorted(dur.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  kernel {k:50s} n={n:4d} total {t / 1e6:.3f} ms avg {t / n / 1e3:.2f} us")
