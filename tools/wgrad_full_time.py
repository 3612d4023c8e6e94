#!/usr/bin/env python
"""Device time of t2v_wgrad_tn at the shapes of full fine-tuning's base-weight gradients (engine_full.py: dW = dy^T x per Linear leaf,
dy^T xcol per conv leaf) on MI355X, inside a hipGraph, by token split (0 = the library's choice), with the result checked against fp32
torch on the first shape.  The output tile is chosen inside the library (128 x 128 where both extents reach 128; T2V_WGRAD_TILE128=0
keeps the 64 x 64 tile): run once per setting, the two CSVs side by side are the A/B.

    T2V_WGRAD_TILE128=0/1 python tools/wgrad_full_time.py [--splits 0,4,8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2v_turbo_amd import native as nt  # noqa: E402
from tools.wgrad_time import graph_us  # noqa: E402

# name, tokens M, R (columns of dy), C (columns of x / xcol), launches per step (about)
SHAPES = [
    ("L0 to_out / proj", 40960, 320, 320, 45),
    ("L0 q|k|v leaf", 40960, 320, 320, 60),
    ("L0 GEGLU proj", 40960, 2560, 320, 10),
    ("L0 ff out", 40960, 320, 1280, 10),
    ("L0 tconv", 40960, 320, 960, 20),
    ("L0 conv3x3", 40960, 320, 2880, 7),
    ("L0 conv3x3 skip", 40960, 320, 5760, 3),
    ("L1 to_out / proj", 10240, 640, 640, 45),
    ("L1 GEGLU proj", 10240, 5120, 640, 10),
    ("L1 ff out", 10240, 640, 2560, 10),
    ("L1 tconv", 10240, 640, 1920, 20),
    ("L1 conv3x3", 10240, 640, 5760, 7),
    ("L1 conv3x3 skip", 10240, 640, 11520, 3),
    ("L2 to_out / proj", 2560, 1280, 1280, 45),
    ("L2 GEGLU proj", 2560, 10240, 1280, 10),
    ("L2 ff out", 2560, 1280, 5120, 10),
    ("L2 tconv", 2560, 1280, 3840, 20),
    ("L2 conv3x3", 2560, 1280, 11520, 7),
    ("L2 conv3x3 skip block", 2560, 1024, 23040, 3),
    ("L3 tconv", 640, 1280, 3840, 28),
    ("L3 conv3x3", 640, 1280, 11520, 8),
    ("L3 conv3x3 skip block", 640, 1024, 23040, 3),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--splits", default="0")
    args = ap.parse_args()
    splits = [int(s) for s in args.splits.split(",")]
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(5)
    ops = nt.HipOps()
    print(f"# T2V_WGRAD_TILE128={os.environ.get('T2V_WGRAD_TILE128', '1')}")
    print("shape,M,R,C,per_step," + ",".join(f"us[splits={s}],tflops" for s in splits) + ",best_ms_per_step")
    total = 0.0
    for i, (name, M, R, C, n) in enumerate(SHAPES):
        a = (torch.randn(M, R, device=dev, generator=gen) * 0.3).bfloat16()
        b = (torch.randn(M, C, device=dev, generator=gen) * 0.3).bfloat16()
        out = torch.zeros(R, C, device=dev)
        cells, best = [], 1e30
        for s in splits:
            try:
                us = graph_us(lambda: ops.wgrad_tn(a, b, out, alpha=1.0, splits=s))
            except Exception as e:   # (a forced split whose partial slabs do not fit the workspace)
                cells += [f"{type(e).__name__}", ""]
                continue
            best = min(best, us)
            cells += [f"{us:.1f}", f"{2.0 * M * R * C / us / 1e6:.0f}"]
        if i == 0:
            ref = a.float().t() @ b.float()
            err = float((out - ref).norm() / ref.norm())
            assert err < 2e-4, err
        total += best * n / 1e3
        print(",".join([name, str(M), str(R), str(C), str(n)] + cells + [f"{best * n / 1e3:.2f}"]), flush=True)
    print(f"# sum over the step's launches (best split per shape): {total:.1f} ms")


if __name__ == "__main__":
    main()
