#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV between two marker kernels (the host brackets the steady-state region with an op
used nowhere else, e.g. ``torch.sinh``): per kernel name calls / total / average inside the window, window wall time and the
GPU-busy fraction.  Writes a small CSV (the raw trace stays on the GPU box).

    python tools/trace_window.py <dir with *_kernel_trace.csv> --marker sinh --out gpurun_out/x/window_stats.csv
"""
import argparse
import csv
import glob
import os
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--marker", default="sinh")
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int, default=1, help="steps inside the window (per-step figures)")
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no kernel trace found"
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]]
    assert len(marks) >= 2, f"need two '{a.marker}' kernels, found {len(marks)}"
    lo, hi = marks[0], marks[-1]
    win = rows[lo + 1:hi]
    t0, t1 = int(rows[lo]["End_Timestamp"]), int(rows[hi]["Start_Timestamp"])
    agg = {}
    busy = 0
    for r in win:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        busy += d
        e = agg.setdefault(short(r["Kernel_Name"]), [0, 0])
        e[0] += 1
        e[1] += d
    with open(a.out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow([f"# window {(t1 - t0) / 1e6:.2f} ms wall, {busy / 1e6:.2f} ms of kernels ({100.0 * busy / max(t1 - t0, 1):.1f} % busy), "
                    f"{len(win)} kernels, {a.steps} step(s)"])
        w.writerow(["kernel", "calls_per_step", "ms_per_step", "avg_us", "pct_of_kernel_time"])
        for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, round(n / a.steps, 1), round(d / 1e6 / a.steps, 3), round(d / 1e3 / n, 2), round(100.0 * d / max(busy, 1), 2)])
    print(open(a.out).read().splitlines()[0])


if __name__ == "__main__":
    main()
