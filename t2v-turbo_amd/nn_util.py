"""Small host-side helpers shared by the module mirrors."""
import math

import torch
import torch.utils.checkpoint


class EngineBox:
    """Holds a lazily built native engine next to an nn.Module without making it part of the module:
    copies / pickles of the module start with an empty box (engines own device buffers)."""

    def __init__(self):
        self.engine = None
        self.enc = None
        self.grad = None
        self.full = None

    def __deepcopy__(self, memo):
        return EngineBox()

    def __reduce__(self):
        return (EngineBox, ())


def checkpoint_call(fn, inputs, flag):
    """Activation checkpointing switch (reference lvdm/common.py:96-112)."""
    if flag:
        return torch.utils.checkpoint.checkpoint(fn, *inputs, use_reentrant=True)
    return fn(*inputs)


def sinusoidal_embedding(timesteps, dim, max_period=10000):
    """cos||sin timestep embedding, fp32 (reference lvdm/models/utils_diffusion.py:8-32)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half).to(timesteps.device)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def guidance_embedding(w, embedding_dim=512, dtype=torch.float32):
    """sin||cos embedding of 1000*w (reference pipeline/t2v_turbo_vc2_pipeline.py:99-120,
    utils/common_utils.py:47-73)."""
    assert len(w.shape) == 1
    w = w * 1000.0
    half = embedding_dim // 2
    emb = torch.log(torch.tensor(10000.0)) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=dtype) * -emb)
    emb = w.to(dtype)[:, None] * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
    if embedding_dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1))
    assert emb.shape == (w.shape[0], embedding_dim)
    return emb


def walk_modules(root):
    """Every module under ``root`` (itself included), depth first, straight off the ``_modules`` dicts: ``nn.Module.modules()``
    builds a name for each of the ~3 000 modules of the UNet on the way (4 ms per walk; this is 1 ms), and the engines walk the
    tree on every call."""
    out, stack = [], [root]
    while stack:
        mod = stack.pop()
        out.append(mod)
        for child in mod._modules.values():
            if child is not None:
                stack.append(child)
    return out


def walk_parameters(root):
    """Every distinct parameter under ``root`` in a fixed (not ``parameters()``'s) order, without building names."""
    out, seen = [], set()
    for mod in walk_modules(root):
        for p in mod._parameters.values():
            if p is not None and id(p) not in seen:
                seen.add(id(p))
                out.append(p)
    return out
