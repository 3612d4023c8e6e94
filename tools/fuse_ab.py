#!/usr/bin/env python
"""Per-launch A/B of the fused normalisation statistics at the UNet's own shapes (MI355X):

    python tools/fuse_ab.py > gpurun_out/fuse_ab.csv

For each level (40960 x 320, 10240 x 640, 2560 x 1280 tokens x channels): the producer of a LayerNorm input (attention
out-projection with residual) with and without row statistics, the standalone LayerNorm, the three consumers (q|k|v, the text
cross-attention's q, the GEGLU projection) on normalised rows and with the LayerNorm folded in; a 3x3 conv and a (3,1,1) conv
with and without column statistics, GroupNorm from the tensor and from the column statistics.  Each line: microseconds per
launch, 40 launches back to back between two events on the launch stream, tuned tiles (gemm_tune.json)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


FORCE_TILES = [int(v) for v in os.environ.get("T2V_AB_TILES", "").split(",") if v]


def main():
    from t2v_turbo_amd import native as nt
    from t2v_turbo_amd.native import HipOps
    ops = HipOps()
    ops.init()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)

    def rnd(*shape, dtype=torch.bfloat16, scale=1.0):
        return (torch.randn(*shape, generator=gen, device=dev) * scale).to(dtype)

    def timeit(fn, n=40):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    print("level,op,variant,us")
    for M, C, frames_rows, hw in ((40960, 320, 2560, 2560), (10240, 640, 640, 640), (2560, 1280, 160, 160)):
        lvl = f"{M}x{C}"
        x, res = rnd(M, C), rnd(M, C)
        wo, bo = rnd(C, C, scale=C ** -0.5), rnd(C, dtype=torch.float32)
        y, ln = torch.empty(M, C, dtype=torch.bfloat16, device=dev), torch.empty(M, C, dtype=torch.bfloat16, device=dev)
        rs = torch.zeros(M, C // 16, device=dev)
        gamma, beta = rnd(C, dtype=torch.float32), rnd(C, dtype=torch.float32)
        print(f"{lvl},out_proj,plain,{timeit(lambda: ops.gemm(x, wo, y, M=M, N=C, bias=bo, residual=res)):.2f}")
        ok = ops.gemm_fuse_supported(x, wo, y, M=M, N=C, bias=bo, residual=res, rowstat=rs)
        print(f"{lvl},out_proj,rowstat{'' if ok else '(unsupported)'},{timeit(lambda: ops.gemm(x, wo, y, M=M, N=C, bias=bo, residual=res, rowstat=rs)) if ok else float('nan'):.2f}")
        print(f"{lvl},layernorm,standalone,{timeit(lambda: ops.layernorm(y, gamma, beta, 1e-5, ln)):.2f}")
        for name, N, act in (("qkv", 3 * C, nt.ACT_NONE), ("cross_q", C, nt.ACT_NONE), ("ff1_geglu", 8 * C, nt.ACT_GEGLU)):
            w = rnd(N, C, scale=C ** -0.5)
            b = rnd(N, dtype=torch.float32)
            s_vec = w.float().sum(1).contiguous()
            out = torch.empty(M, N // 2 if act == nt.ACT_GEGLU else N, dtype=torch.bfloat16, device=dev)
            print(f"{lvl},{name},plain,{timeit(lambda: ops.gemm(ln, w, out, M=M, N=N, bias=b, act=act)):.2f}")
            kw = dict(M=M, N=N, bias=b, act=act, lnf=(rs, 1e-5, s_vec))
            ok = ops.gemm_fuse_supported(y, w, out, **kw)
            print(f"{lvl},{name},ln_fold{'' if ok else '(unsupported)'},{timeit(lambda: ops.gemm(y, w, out, **kw)) if ok else float('nan'):.2f}")
            for cfg in FORCE_TILES:   # the fold on other tiles than the tuned one (the tuning was done for the plain launch)
                if act == nt.ACT_GEGLU and cfg in (5, 9, 23, 31):
                    continue
                kw2 = dict(kw, tile_cfg=cfg, split_k=1)
                if ops.gemm_fuse_supported(y, w, out, **kw2):
                    print(f"{lvl},{name},ln_fold@tile{cfg},{timeit(lambda: ops.gemm(y, w, out, **kw2)):.2f}")
        # the feed-forward: LayerNorm + GEGLU projection + output projection (+ residual) as three launches vs t2v_ffn_fused
        if ops.ffn_fused_supported(C):
            w1, b1 = rnd(8 * C, C, scale=C ** -0.5), rnd(8 * C, dtype=torch.float32)
            w2, b2 = rnd(C, 4 * C, scale=(4 * C) ** -0.5), rnd(C, dtype=torch.float32)
            hbuf, o3 = torch.empty(M, 4 * C, dtype=torch.bfloat16, device=dev), torch.empty(M, C, dtype=torch.bfloat16, device=dev)
            pk = nt.ffn_pack(w1, b1, w2, b2, gamma, beta, torch.bfloat16)

            def three():
                ops.layernorm(y, gamma, beta, 1e-5, ln)
                ops.gemm(ln, w1, hbuf, M=M, N=8 * C, bias=b1, act=nt.ACT_GEGLU)
                ops.gemm(hbuf, w2, o3, M=M, N=C, bias=b2, residual=y)
            print(f"{lvl},feed_forward,three_launches,{timeit(three):.2f}")
            print(f"{lvl},feed_forward,fused,{timeit(lambda: ops.ffn_fused(y, *pk, 1e-5, o3)):.2f}")
        # GroupNorm producers / consumers
        n_img, h, wd = 16, {2560: 40, 640: 20, 160: 10}[hw], {2560: 64, 640: 32, 160: 16}[hw]
        w3, w1 = rnd(C, 9 * C, scale=(9 * C) ** -0.5), rnd(C, 3 * C, scale=(3 * C) ** -0.5)
        out = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
        cs = torch.zeros(M // 32, C, 2, device=dev)
        for name, w, kw in (("conv3x3", w3, dict(mode=nt.GEMM_CONV3X3, n_img=n_img, h=h, wd=wd)),
                            ("tconv3", w1, dict(mode=nt.GEMM_TCONV3, n_img=n_img, h=h, wd=wd, frames=16))):
            print(f"{lvl},{name},plain,{timeit(lambda: ops.gemm(x, w, out, M=M, N=C, bias=bo, residual=res, **kw)):.2f}")
            ok = ops.gemm_fuse_supported(x, w, out, M=M, N=C, bias=bo, residual=res, colstat=cs, **kw)
            print(f"{lvl},{name},colstat{'' if ok else '(unsupported)'},{timeit(lambda: ops.gemm(x, w, out, M=M, N=C, bias=bo, residual=res, colstat=cs, **kw)) if ok else float('nan'):.2f}")
        for units, rows in ((16, frames_rows), (1, M)):
            ws = torch.zeros(max(ops.group_norm_ws_floats(units, rows, 32, C), ops.group_norm_cs_ws_floats(units, rows, 32), 1), device=dev)
            print(f"{lvl},group_norm[{units}x{rows}],tensor,{timeit(lambda: ops.group_norm(out, None, units, rows, 1e-5, gamma, beta, True, ws, ln)):.2f}")
            print(f"{lvl},group_norm[{units}x{rows}],colstat,{timeit(lambda: ops.group_norm_cs(cs, None, out, None, units, rows, 1e-5, gamma, beta, True, ws, ln)):.2f}")


if __name__ == "__main__":
    main()
