#!/usr/bin/env python
"""Time the flash attention kernel on one shape (ablation bits via --debug)."""
import argparse, ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2v_turbo_amd import native as nt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nimg", type=int, default=16)
ap.add_argument("--seq", type=int, default=2560)
ap.add_argument("--kv", type=int, default=0)
ap.add_argument("--heads", type=int, default=5)
ap.add_argument("--debug", type=int, default=0)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
ops = nt.HipOps(); ops.init()
ops.lib.t2v_attn_debug.argtypes = [ctypes.c_int]
ops.lib.t2v_attn_debug(a.debug)
kv = a.kv or a.seq
inner = a.heads * 64
kp = (kv + 63) // 64 * 64
torch.manual_seed(0)
q = torch.randn(a.nimg * a.seq, inner, device="cuda").bfloat16()
k = torch.randn(a.nimg * kv, inner, device="cuda").bfloat16()
vt = torch.randn(a.nimg * inner, kp, device="cuda").bfloat16()
out = torch.empty_like(q)
for _ in range(2):
    ops.attn_spatial(q, k, vt, kp, out, a.nimg, a.seq, kv, a.heads, 1, 0.125)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    ops.attn_spatial(q, k, vt, kp, out, a.nimg, a.seq, kv, a.heads, 1, 0.125)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / a.iters
chk = int(out.view(torch.int16).to(torch.int64).mul(torch.arange(out.numel(), device="cuda").view_as(out) % 8191 + 1).sum().item())
print(f"attn debug={a.debug} nimg={a.nimg} seq={a.seq} kv={kv} heads={a.heads}: {us:.1f} us  {4.0*a.nimg*a.heads*a.seq*kv*64/us/1e6:.1f} TF/s  checksum {chk}")
