#!/usr/bin/env python
"""One-off CPU check of the native student's dataflow at FULL WIDTH (VideoCrafter2 channel counts 320..1280, rank 64, 575
injected leaves, 1150 LoRA tensors) on a small latent: every LoRA gradient + d/d(latents) of the emulated engine against
torch autograd.  The unit tests run the same comparison at 64 channels; this covers what only the real widths exercise
(two-part 2560-channel inputs, 10240-row GEGLU permutation, arena sizes).  ~10 min and ~25 GB of host memory.

    python tools/check_lora_grad_fullwidth.py [F H W]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from t2v_turbo_amd import lora
    from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
    from t2v_turbo_amd.unet3d import UNetModel
    from tests.emu_ops import EmuOps
    from tests.util import VC2_UNET, rel_l2
    F, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (2, 16, 16)
    t0 = time.time()
    torch.manual_seed(0)
    m = UNetModel(**VC2_UNET).eval()
    gen = torch.Generator().manual_seed(1234)
    with torch.no_grad():
        for p in m.parameters():
            if float(p.abs().max()) == 0.0:
                p.normal_(0.0, 0.02, generator=gen)
    m.requires_grad_(False)
    lora.inject_trainable_lora_extended(m, r=64)
    params = lora.lora_parameters(m)
    with torch.no_grad():
        for p in params:
            p.copy_(torch.randn(p.shape, generator=gen) * 0.02)
    m.eval()
    print(f"built: {len(params)} LoRA tensors, {sum(p.numel() for p in params) / 1e6:.1f} M elements ({time.time() - t0:.0f}s)", flush=True)
    x = torch.randn(1, 4, F, H, W, generator=gen)
    ts = torch.tensor([519])
    ctx = torch.randn(1, 77, 1024, generator=gen)
    tc = torch.randn(1, 256, generator=gen)
    r_out = torch.randn(x.shape, generator=gen)
    m.native_mode = "off"
    xg = x.clone().requires_grad_(True)
    y_ref = m(xg, ts, context=ctx, fps=16, timestep_cond=tc)
    (y_ref * r_out).sum().backward()
    g_ref = [p.grad.clone() for p in params]
    dx_ref = xg.grad.clone()
    print(f"autograd done ({time.time() - t0:.0f}s)", flush=True)
    eng = UNetGradEngine(m, EmuOps(strict=True))
    eng.bind_lora(params)
    emb_all = m.conditioning_emb_all(ts, 16, tc, None)
    y = eng.forward_tape(x, ts, ctx, 16, tc, None, emb_all=emb_all)
    flat = torch.zeros(eng.lora_numel)
    dx = eng.backward(r_out, flat_grad=flat, accumulate=False)
    for p in params:
        p.grad = None
    emb_all.backward(eng.d_emb_all)
    print(f"engine done ({time.time() - t0:.0f}s); operand arena {eng.lp_used / 1e6:.1f} M bf16, gradient arena {eng.e_used / 1e6:.1f} M fp32",
          flush=True)
    names = {id(p): n for n, p in m.named_parameters()}
    worst, off, zero = (0.0, None), 0, 0
    for p, r in zip(params, g_ref):
        g = flat[off:off + p.numel()].view_as(p)
        if p.grad is not None:
            g = g + p.grad
        off += p.numel()
        if float(r.abs().max()) == 0:
            zero += 1
            assert float(g.abs().max()) < 1e-7, names[id(p)]
            continue
        e = rel_l2(g, r)
        if e > worst[0]:
            worst = (e, names[id(p)])
    print(f"out rel-L2 {rel_l2(y, y_ref.detach()):.2e}  dx rel-L2 {rel_l2(dx, dx_ref):.2e}  worst LoRA gradient {worst[0]:.2e} ({worst[1]}), "
          f"{zero} all-zero reference gradients")
    assert rel_l2(y, y_ref.detach()) < 1e-4 and rel_l2(dx, dx_ref) < 1e-3 and worst[0] < 1e-3
    print("OK")


if __name__ == "__main__":
    main()
