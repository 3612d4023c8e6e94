"""Statistics of the counter-based dropout mask (csrc/common.h: one splitmix64 word per quad of adjacent elements, 16 bits per
element; mirrored bit for bit by tests/emu_ops.EmuOps.dropout_keep, which the simulator and GPU suites compare the kernels with).
The reference draws its masks from torch's generator (nn.Dropout in utils/lora.py:45-50 and openaimodel3d.py:280-297); ours is a
different stream with the same law: independent Bernoulli(1 - p) keeps.  This file checks the law."""
import math

import pytest
import torch

from tests.emu_ops import EmuOps


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_rate_and_independence(p):
    rows, ncols = 512, 1024
    keep = EmuOps.dropout_keep(0x5EED_0001, 3, rows, ncols, p).float()
    n = rows * ncols
    q = 1.0 - (int(p * 4294967296.0) >> 16) / 65536.0          # the law actually drawn: p resolved to 2^-16
    assert abs(q - (1 - p)) < 2e-5
    sd = math.sqrt(q * (1 - q) / n)
    assert abs(float(keep.mean()) - q) < 5 * sd
    drop = 1.0 - keep
    # neighbours inside a quad (same 64-bit word), across quads, across rows, and the four slots of a word: no correlation
    for a, b in ((drop[:, :-1], drop[:, 1:]), (drop[:, :-4], drop[:, 4:]), (drop[:-1], drop[1:])):
        joint = float((a * b).mean())
        assert abs(joint - (1 - q) ** 2) < 6 * math.sqrt((1 - q) ** 2 / a.numel())
    slots = drop.reshape(rows, ncols // 4, 4).mean(dim=(0, 1))
    assert float((slots - (1 - q)).abs().max()) < 6 * math.sqrt(q * (1 - q) / (n / 4))


def test_sites_and_seeds_give_unrelated_masks():
    a = EmuOps.dropout_keep(1, 3, 256, 256, 0.5)
    b = EmuOps.dropout_keep(1, 4, 256, 256, 0.5)
    c = EmuOps.dropout_keep(2, 3, 256, 256, 0.5)
    for other in (b, c):
        agree = float((a == other).float().mean())
        assert abs(agree - 0.5) < 0.02
    assert torch.equal(a, EmuOps.dropout_keep(1, 3, 256, 256, 0.5))   # a pure function of (seed, site, position)
