#!/usr/bin/env python
"""One shape of t2v_ffn_fused in a loop (for rocprofv3 --pmc passes and quick timings):
    python tools/ffn_one.py --m 40960 --c 320 --iters 10"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=40960)
    ap.add_argument("--c", type=int, default=320)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    from t2v_turbo_amd import native as nt
    ops = nt.HipOps()
    ops.init()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    M, C = a.m, a.c
    rnd = lambda *s, sc=1.0: torch.randn(*s, generator=g, device=dev) * sc   # noqa: E731
    x = rnd(M, C).bfloat16()
    pk = nt.ffn_pack(rnd(8 * C, C, sc=C ** -0.5), rnd(8 * C, sc=0.1), rnd(C, 4 * C, sc=(4 * C) ** -0.5), rnd(C, sc=0.1), rnd(C) * 0.2 + 1, rnd(C) * 0.1,
                     torch.bfloat16)
    out = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        ops.ffn_fused(x, *pk, 1e-5, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        ops.ffn_fused(x, *pk, 1e-5, out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    tf = 2.0 * M * 12 * C * C / 1e12
    print(f"t2v_ffn_fused M={M} C={C} CT={os.environ.get('T2V_FFN_CT', '3')}: {us:.1f} us  {tf / us * 1e6:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
