"""Deterministic synthetic weights / inputs (oracle, test-only).

A freshly constructed reference UNet outputs exactly zero (``zero_module``,
reference ``lvdm/basics.py:20-26``), so parity on default init is vacuous
(SURVEY.md §0.4).  Tests, goldens and the bench's CPU baseline therefore use a
synthetic state dict drawn from a fixed, reference-independent recipe: each
tensor is seeded by the CRC32 of its key, so the reference model (in
``tests/golden/make_golden.py``), the oracle and the device under test all see
bit-identical fp32 weights without shipping them.
"""
import math
import zlib

import torch


def synth_tensor(key, shape, seed=1234):
    g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
    t = torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    if len(shape) == 1:
        if key.endswith(".weight"):  # norm gains
            return 1.0 + 0.1 * t
        return 0.05 * t  # biases
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return t / math.sqrt(fan_in)


def synth_state_dict(manifest, seed=1234):
    """manifest: ordered list of [key, shape]."""
    return {k: synth_tensor(k, s, seed) for k, s in manifest}


def manifest_of(module):
    return [[k, list(v.shape)] for k, v in module.state_dict().items()]


def synth_inputs(b, frames, h, w, ctx_len=77, ctx_dim=1024, cond_dim=256, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, 4, frames, h, w, generator=g)
    ctx = torch.randn(b, ctx_len, ctx_dim, generator=g)
    return x, ctx
