#!/usr/bin/env python
"""Run one GEMM shape repeatedly (for rocprofv3 counter collection / quick timing).
    python tools/gemm_one.py --mode 1 --nimg 16 --h 20 --w 32 --cin 640 --n 640 --cfg 1 --iters 20"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2v_turbo_amd import native as nt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", type=int, default=0)
    ap.add_argument("--m", type=int, default=0)
    ap.add_argument("--nimg", type=int, default=16)
    ap.add_argument("--h", type=int, default=20)
    ap.add_argument("--w", type=int, default=32)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--cin", type=int, default=640)
    ap.add_argument("--n", type=int, default=640)
    ap.add_argument("--cfg", type=int, default=0)
    ap.add_argument("--split", type=int, default=0)
    ap.add_argument("--act", type=int, default=0)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--debug", type=int, default=0, help="ablation bits; need a -DT2V_GEMM_ABLATE build (T2V_EXTRA_HIPCC_FLAGS)")
    ap.add_argument("--res", type=int, default=0, help="add a residual input")
    ap.add_argument("--graph", type=int, default=1, help="time inside a hipGraph")
    a = ap.parse_args()
    ops = nt.HipOps()
    ops.init()
    import ctypes
    ops.lib.t2v_gemm_debug.argtypes = [ctypes.c_int]
    ops.lib.t2v_gemm_debug(a.debug)
    taps = {0: 1, 4: 3}.get(a.mode, 9)
    rows = a.m if a.mode == 0 else a.nimg * a.h * a.w
    M = rows if a.mode in (0, 1, 4) else (rows // 4 if a.mode in (2, 5) else rows * 4)
    x = torch.randn(rows, a.cin, device="cuda").bfloat16()
    wt = (torch.randn(a.n, taps * a.cin, device="cuda") * (taps * a.cin) ** -0.5).bfloat16()
    bias = torch.randn(a.n, device="cuda")
    out = torch.empty(M, a.n // 2 if a.act == 1 else a.n, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(M, a.n, device="cuda").bfloat16() if a.res else None
    kw = dict(M=M, N=a.n, mode=a.mode, n_img=a.nimg, h=a.h, wd=a.w, frames=a.frames, bias=bias, act=a.act,
              tile_cfg=a.cfg, split_k=a.split, residual=res)
    for _ in range(3):
        ops.gemm(x, wt, out, **kw)
    torch.cuda.synchronize()
    if a.graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(a.iters):
                ops.gemm(x, wt, out, **kw)
        g.replay()
        torch.cuda.synchronize()
    us = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if a.graph:
            g.replay()
        else:
            for _ in range(a.iters):
                ops.gemm(x, wt, out, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = min(us, e0.elapsed_time(e1) * 1e3 / a.iters)
    print(f"debug={a.debug} mode={a.mode} M={M} N={a.n} K={taps * a.cin} cfg={a.cfg} split={a.split}: {us:.1f} us  "
          f"{2.0 * M * a.n * taps * a.cin / us / 1e6:.1f} TF/s")


if __name__ == "__main__":
    main()
