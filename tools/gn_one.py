#!/usr/bin/env python
"""Run t2v_group_norm on a few shapes repeatedly (for rocprofv3 --kernel-trace --stats per-kernel durations).
    python tools/gn_one.py --shapes 320:1:40960,1280:16:40 --iters 50"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2v_turbo_amd import native as nt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="320:1:40960")
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
ops = nt.HipOps()
ops.init()
for sh in a.shapes.split(","):
    C, units, rows = (int(v) for v in sh.split(":"))
    x = torch.randn(units * rows, C, device="cuda").bfloat16()
    out = torch.empty_like(x)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    ws = torch.empty(ops.group_norm_ws_floats(units, rows, 32, C), device="cuda")
    for _ in range(a.iters):
        ops.group_norm(x, None, units, rows, 1e-5, g, b, True, ws, out)
    torch.cuda.synchronize()
    print("done", sh)
