#!/usr/bin/env python
"""Golden FULL fine-tuning gradients from THE REFERENCE ITSELF (CPU, build container only; needs /root/reference):

    python tests/golden/make_golden_full_grad.py               -> tests/golden/unet_tiny_full_grad.npz   (model_channels 64)
    python tests/golden/make_golden_full_grad.py --width 128   -> tests/golden/unet_mid_full_grad.npz    (model_channels 128: 128-512-channel
                                                                  levels, 2 / 4 / 8 heads; digests only)

The student's backward of train_latent_t2v_turbo_v2.py (:669 ``unet.requires_grad_(True)``, :798-816 every parameter in an optimizer
group, :1262 backward) at tiny width: the reference ``UNetModel`` with every parameter trainable, ``eval()`` (the temporal-conv dropouts
are the one thing the native path does not reproduce bit for bit: counter-based masks), loss = <output, r_out>.  Stored: the output,
d(loss)/d(latents), and per parameter (``named_parameters`` order, names included) three numbers — L2 norm and two seeded random
projections of its gradient (any layout / permutation / scale slip moves them) — plus the full gradients of a few small parameters (the
4-channel entry / exit convs, one GroupNorm, one LayerNorm, one bias of every kind).  Inputs are those of ``unet_tiny.npz``; weights come
from ``oracle/synth.py`` (nothing is zero-initialised there)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

SEED_R, SEED_P = 9, 4321
KEEP_FULL = ("input_blocks.0.0.weight", "input_blocks.0.0.bias", "out.2.weight", "out.2.bias", "out.0.weight", "out.0.bias",
             "input_blocks.1.1.transformer_blocks.0.norm2.weight", "input_blocks.1.1.transformer_blocks.0.norm2.bias",
             "input_blocks.1.1.transformer_blocks.0.ff.net.0.proj.bias", "input_blocks.1.0.temopral_conv.conv2.3.bias")


def digests(grads):
    gen = torch.Generator().manual_seed(SEED_P)
    out = []
    for g in grads:
        f = g.detach().double().reshape(-1)
        v1 = torch.randn(f.numel(), generator=gen, dtype=torch.float64)
        v2 = torch.randn(f.numel(), generator=gen, dtype=torch.float64)
        out.append([float(f.norm()), float(f @ v1), float(f @ v2)])
    return np.asarray(out, dtype=np.float64)


def main():
    width = int(sys.argv[sys.argv.index("--width") + 1]) if "--width" in sys.argv else 64
    import make_golden as mg
    mg.install_stubs()
    sys.path.insert(0, mg.REF)
    from oracle.synth import manifest_of, synth_state_dict
    from lvdm.modules.networks.openaimodel3d import UNetModel

    z = np.load(os.path.join(HERE, "unet_tiny.npz"))
    x, ts, ctx, tc = (torch.from_numpy(z[k]) for k in ("x", "ts", "ctx", "tc"))
    m = UNetModel(**mg.tiny_unet_params(model_channels=width))
    m.load_state_dict(synth_state_dict(manifest_of(m)), strict=True)
    m.requires_grad_(True)
    m.eval()
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(SEED_R))
    xg = x.clone().requires_grad_(True)
    out = m(xg, ts, context=ctx, fps=16, timestep_cond=tc)
    (out * r_out).sum().backward()
    named = list(m.named_parameters())
    assert all(p.grad is not None for _, p in named)
    full = {"g_" + n.replace(".", "__"): p.grad.numpy() for n, p in named if n in KEEP_FULL}
    assert len(full) == len(KEEP_FULL), sorted(set(KEEP_FULL) - {n for n, _ in named})
    if width != 64:
        full = {}          # (the second anchor keeps the digests only)
    np.savez_compressed(os.path.join(HERE, "unet_tiny_full_grad.npz" if width == 64 else "unet_mid_full_grad.npz"), width=np.int64(width), out=out.detach().numpy(), dx=xg.grad.numpy(), r_out=r_out.numpy(),
                        digests=digests([p.grad for _, p in named]), names=np.asarray([n for n, _ in named]), **full)
    print(len(named), "parameters,", sum(p.numel() for _, p in named), "elements; full gradients kept for", len(full))


if __name__ == "__main__":
    main()
