#!/usr/bin/env python
"""Per-launch time of the two attention backward entry points at the student UNet's shapes (MI355X):

    python tools/attn_bwd_ab.py                         # product library, MFMA temporal backward
    T2V_TATTN_BWD_VALU=1 python tools/attn_bwd_ab.py    # the first (VALU / LDS) temporal backward
    T2V_HIP_LIB=<variant .so> python tools/attn_bwd_ab.py

One CSV line per (op, shape): microseconds per launch (20 launches between two events on the launch stream), the bytes the
launch has to move and the rate that gives.  Temporal: (clips*hw tokens) x heads problems of 16 frames; spatial: 16 frames x
(hw tokens) self-attention, both at head dim 64 (attention.py:331-384 under autograd in the reference)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from t2v_turbo_amd.native import HipOps
    ops = HipOps()
    ops.init()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    tag = os.environ.get("T2V_AB_TAG", "valu" if os.environ.get("T2V_TATTN_BWD_VALU") == "1" else "default")

    def rnd(*shape):
        return torch.randn(*shape, generator=gen, device=dev).bfloat16()

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    print("build,op,shape,us,mbytes,gb_per_s")
    F = 16
    for hw, heads in ((2560, 5), (2560, 8), (640, 10), (160, 20), (40, 20)):
        M, inner = F * hw, heads * 64
        qkv, do = rnd(M, 3 * inner), rnd(M, inner)
        g = torch.empty(M, 3 * inner, dtype=torch.bfloat16, device=dev)
        us = timeit(lambda: ops.attn_temporal_bwd(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], do, None, g[:, :inner],
                                                  g[:, inner:2 * inner], g[:, 2 * inner:], 1, F, hw, heads, 0.125))
        mb = 7 * M * inner * 2 / 1e6
        print(f"{tag},attn_temporal_bwd,{hw}x{heads},{us:.1f},{mb:.1f},{mb / us * 1e3:.0f}")
    for seq, heads in ((2560, 5), (640, 10), (160, 20)):
        n_img, inner = F, heads * 64
        M = n_img * seq
        sp = (seq + 63) // 64 * 64
        qkv, do, o = rnd(M, 3 * inner), rnd(M, inner), rnd(M, inner)
        q, k, v = qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:]

        def tp(x):
            out = torch.zeros(n_img * inner, sp, dtype=torch.bfloat16, device=dev)
            out.view(n_img, inner, sp)[:, :, :seq] = x.view(n_img, seq, inner).transpose(1, 2)
            return out
        kt, qt, dot = tp(k), tp(q), tp(do)
        l2, ds = torch.zeros(n_img * heads, sp, device=dev), torch.zeros(n_img * heads, sp, device=dev)
        g = torch.empty(M, 3 * inner, dtype=torch.bfloat16, device=dev)
        us = timeit(lambda: ops.attn_spatial_bwd(q, k, v, seq * v.stride(0), 64, kt, qt, dot, do, o, l2, ds, g[:, :inner], g[:, inner:2 * inner],
                                                 g[:, 2 * inner:], n_img, seq, heads, 0.125), n=10)
        flop = 2 * n_img * heads * seq * seq * 64 * (3 + 2 + 4 + 1)   # dq: scores x2 + dP + dQ; dkv: S, dP, dV, dK
        print(f"{tag},attn_spatial_bwd,{seq}x{heads},{us:.1f},{flop / 1e9:.1f} GFLOP,{flop / us / 1e6:.0f} TFLOP/s")


if __name__ == "__main__":
    main()
