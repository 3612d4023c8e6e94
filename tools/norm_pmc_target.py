#!/usr/bin/env python
"""Target for rocprofv3 passes over the normalisation kernels of the UNet step at their in-step shapes (VERDICT r4 "next" 3a):
GroupNorm(+SiLU) on the producers' column statistics (gn_partial_cs_kernel + gn_apply_kernel<2>) at the three UNet levels, and the
LayerNorm of a spatial transformer at 40 960 x 320 (layernorm_rows_kernel).
    python tools/norm_pmc_target.py --iters 10"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2v_turbo_amd import native as nt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--gn", default="320:16:2560,640:16:640,1280:16:160")
ap.add_argument("--ln", default="40960:320,10240:640")
a = ap.parse_args()
ops = nt.HipOps()
ops.init()
torch.manual_seed(0)
for sh in a.gn.split(","):
    C, units, rows = (int(v) for v in sh.split(":"))
    x = torch.randn(units * rows, C, device="cuda").bfloat16()
    xf = x.float().view(units * rows // 32, 32, C)
    cs = torch.stack([xf.sum(1), (xf * xf).sum(1)], dim=-1).contiguous()          # [slab of 32 rows][channel][2]
    out = torch.empty_like(x)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    ws = torch.empty(ops.group_norm_cs_ws_floats(units, rows, 32), device="cuda")
    for _ in range(a.iters):
        ops.group_norm_cs(cs, None, x, None, units, rows, 1e-5, g, b, True, ws, out)
    torch.cuda.synchronize()
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(
        x.float().view(units, rows, C).transpose(1, 2), 32, g, b, 1e-5)).transpose(1, 2).reshape(units * rows, C)
    print("gn", sh, "max err", float((out.float() - ref).abs().max()))
for sh in a.ln.split(","):
    M, C = (int(v) for v in sh.split(":"))
    x = torch.randn(M, C, device="cuda").bfloat16()
    out = torch.empty_like(x)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    for _ in range(a.iters):
        ops.layernorm(x, g, b, 1e-5, out)
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x.float(), (C,), g, b, 1e-5)
    print("ln", sh, "max err", float((out.float() - ref).abs().max()))
