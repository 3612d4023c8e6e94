#!/usr/bin/env python
"""Golden gradients from THE REFERENCE ITSELF (CPU, build container only; needs /root/reference):

    python tests/golden/make_golden_grad.py        -> tests/golden/unet_tiny_grad.npz

What the UNet data-gradient engine must reproduce: autograd through the reference ``UNetModel`` (a) for a linear
functional of the denoiser output and (b) for the motion-prior loss of ``utils/common_utils.py:446-478`` on the recorded
temporal attention probabilities, exactly as ``motion_prior_sample.py:59-84`` computes its score.  Inputs are those of
``unet_tiny.npz``; weights come from ``oracle/synth.py``."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)


def main():
    import make_golden as mg
    mg.install_stubs()
    sys.path.insert(0, mg.REF)
    from oracle.synth import manifest_of, synth_state_dict
    from lvdm.modules.networks.openaimodel3d import UNetModel
    from utils.common_utils import compute_temp_loss          # the reference's own loss

    z = np.load(os.path.join(HERE, "unet_tiny.npz"))
    x, ts, ctx, tc = (torch.from_numpy(z[k]) for k in ("x", "ts", "ctx", "tc"))
    m = UNetModel(**mg.tiny_unet_params(record_attn_probs=True)).eval()
    m.load_state_dict(synth_state_dict(manifest_of(m)), strict=True)
    m.requires_grad_(False)
    g = torch.Generator().manual_seed(21)
    r_out = torch.randn(x.shape, generator=g)
    example = torch.randn(x.shape, generator=g)
    kw = dict(context=ctx, fps=16, timestep_cond=tc)

    def probs_of(model):  # motion_prior_sample.py:40-56
        blocks = tuple(f"output_blocks.{i}.2" for i in range(3, 12))
        return {n: mod.attention_probs for n, mod in model.named_modules()
                if n.startswith(blocks) and n.endswith("blocks.0.attn1")}

    # (a) linear functional of the output
    xg = x.clone().requires_grad_(True)
    (dx_out,) = torch.autograd.grad((m(xg, ts, **kw) * r_out).sum(), xg)
    # (b) the motion-prior score (motion_prior_sample.py:59-84), temp_loss_scale = 500
    with torch.no_grad():
        m(example, ts, **kw)
        ref_probs = {k: v.clone() for k, v in probs_of(m).items()}
    xg = x.clone().requires_grad_(True)
    out = m(xg, ts, **kw)
    loss = 500.0 * compute_temp_loss(probs_of(m), ref_probs)
    (score,) = torch.autograd.grad(loss, xg)
    assert len(ref_probs) == 9 and float(score.abs().max()) > 0
    np.savez_compressed(os.path.join(HERE, "unet_tiny_grad.npz"), r_out=r_out.numpy(), example=example.numpy(),
                        dx_out=dx_out.numpy(), score=score.numpy(), loss=np.float32(loss.item()), out=out.detach().numpy())
    print("dx_out std", float(dx_out.std()), "score std", float(score.std()), "loss", float(loss))


if __name__ == "__main__":
    main()
