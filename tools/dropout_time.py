#!/usr/bin/env python
"""Device time of t2v_dropout_bf16 at the student's shapes (MI355X): rows x ncols bf16, read + write, with the achieved traffic."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2v_turbo_amd.native import HipOps  # noqa: E402


def main():
    ops = HipOps()
    dev = torch.device("cuda", 0)
    seed = torch.tensor([1234567], dtype=torch.int64, device=dev)
    print("rows,ncols,resid,us,tb_per_s")
    for rows, ncols, resid in ((40960, 320, False), (40960, 320, True), (40960, 2560, False), (10240, 640, False), (10240, 5120, False),
                               (2560, 1280, False), (2560, 10240, False), (640, 1280, False)):
        x = torch.randn(rows, ncols, device=dev).bfloat16()
        r = torch.randn(rows, ncols, device=dev).bfloat16() if resid else None
        o = torch.empty_like(x)
        for _ in range(3):
            ops.dropout(x, r, o, ncols, 0.1, seed, 5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.dropout(x, r, o, ncols, 0.1, seed, 5)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        byts = rows * ncols * 2 * (3 if resid else 2)
        print(f"{rows},{ncols},{int(resid)},{us:.2f},{byts / us / 1e6:.2f}")


if __name__ == "__main__":
    main()
