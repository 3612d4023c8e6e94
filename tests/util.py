"""Shared test helpers (configs of the golden cases, fixture loading, error metrics)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# UNet kwargs of configs/inference_t2v_512_v2.0.yaml:25-50 (+ the two keys every t2v-turbo
# entry point sets: use_checkpoint=False for inference, time_cond_proj_dim=256)
VC2_UNET = dict(
    in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
    num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_head_channels=64, transformer_depth=1,
    context_dim=1024, use_linear=True, use_checkpoint=False, temporal_conv=True,
    temporal_attention=True, temporal_selfatt_only=True, use_relative_position=False,
    use_causal_attention=False, temporal_length=16, addition_attention=True, fps_cond=True,
    time_cond_proj_dim=256,
)


def tiny_unet_params(**over):
    p = dict(VC2_UNET)
    p.update(model_channels=64, context_dim=128)
    p.update(over)
    return p


VAE_TINY_DD = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=64,
                   ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
VAE_FULL_DD = dict(double_z=True, z_channels=4, resolution=512, in_channels=3, out_ch=3, ch=128,
                   ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


_MAN = None


def manifest(name):
    global _MAN
    if _MAN is None:
        _MAN = json.load(open(os.path.join(GOLDEN, "manifests.json")))
    return _MAN[name]


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
