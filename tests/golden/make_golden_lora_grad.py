#!/usr/bin/env python
"""Golden LoRA gradients from THE REFERENCE ITSELF (CPU, build container only; needs /root/reference):

    python tests/golden/make_golden_lora_grad.py              -> tests/golden/unet_tiny_lora_grad.npz   (model_channels 64)
    python tests/golden/make_golden_lora_grad.py --width 128  -> tests/golden/unet_mid_lora_grad.npz    (model_channels 128: a second,
                                                                 wider anchor for the full-width chain, whose CPU reference is the
                                                                 repository's own composite module: oracle/lora_grad_oracle.py)

The student's backward of the distillation step (train_t2v_turbo_v1_lora.py:1190) at tiny width: the reference
``UNetModel`` with the reference's own ``utils.lora.inject_trainable_lora_extended`` (rank 64), both LoRA factors drawn
from a seeded generator in injection order, ``eval()`` (the LoRA / temporal-conv dropouts are the one thing the native
path does not reproduce), loss = <output, r_out>.  Stored: the output, d(loss)/d(latents), and per LoRA tensor three
numbers — L2 norm and two seeded random projections of its gradient (any layout / permutation / scale slip moves them) —
plus the full gradients of the two rank-4 leaves (entry / exit conv).  Inputs are those of ``unet_tiny.npz``."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

SEED_W, SEED_R, SEED_P = 7, 5, 1234


def draw_lora(params):
    gen = torch.Generator().manual_seed(SEED_W)
    with torch.no_grad():
        for p in params:
            p.copy_(torch.randn(p.shape, generator=gen) * 0.05)


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
    from utils.lora import inject_trainable_lora_extended    # the reference's own injection

    z = np.load(os.path.join(HERE, "unet_tiny.npz"))
    x, ts, ctx, tc = (torch.from_numpy(z[k]) for k in ("x", "ts", "ctx", "tc"))
    m = UNetModel(**mg.tiny_unet_params(model_channels=width)).eval()
    m.load_state_dict(synth_state_dict(manifest_of(m)), strict=True)
    m.requires_grad_(False)
    its, names = inject_trainable_lora_extended(m, target_replace_module={"UNetModel"}, r=64)  # train_t2v_turbo_v1_lora.py:644-657
    params = [p for it in its for p in it]                     # [up0, down0, up1, down1, ...]
    draw_lora(params)
    m.eval()
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(SEED_R))
    xg = x.clone().requires_grad_(True)
    out = m(xg, ts, context=ctx, fps=16, timestep_cond=tc)
    (out * r_out).sum().backward()
    grads = [p.grad for p in params]
    shapes = np.asarray([list(p.shape) + [0] * (5 - p.dim()) for p in params], dtype=np.int64)
    small = {f"g{i}": g.numpy() for i, g in enumerate(grads) if g.numel() <= 4 * 64 * 9 and min(g.shape[:2]) == 4}
    name = "unet_tiny_lora_grad.npz" if width == 64 else "unet_mid_lora_grad.npz"
    if width != 64:
        small = {}          # (the second anchor keeps the digests only)
    np.savez_compressed(os.path.join(HERE, name), out=out.detach().numpy(), dx=xg.grad.numpy(),
                        digests=digests(grads), shapes=shapes, n_leaves=np.int64(len(names)), width=np.int64(width), **small)
    print(len(params), "LoRA tensors,", sum(p.numel() for p in params), "elements; full gradients kept for", sorted(small))


if __name__ == "__main__":
    main()
