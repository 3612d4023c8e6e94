#!/usr/bin/env python
"""Target of the rocprofv3 --pmc passes over t2v_wgrad_tn at three base-weight gradient shapes of full fine-tuning (the 640-channel 3x3 conv,
the 1 280-channel 3x3 conv, the 640-channel GEGLU projection), eager, 5 launches each; T2V_WGRAD_TILE128=0/1 selects the output tile."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2v_turbo_amd import native as nt  # noqa: E402

ops = nt.HipOps()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(3)
for M, R, C in ((10240, 640, 5760), (2560, 1280, 11520), (10240, 5120, 640)):
    a = (torch.randn(M, R, device=dev, generator=gen) * 0.3).bfloat16()
    b = (torch.randn(M, C, device=dev, generator=gen) * 0.3).bfloat16()
    out = torch.zeros(R, C, device=dev)
    for _ in range(5):
        ops.wgrad_tn(a, b, out)
    torch.cuda.synchronize()
