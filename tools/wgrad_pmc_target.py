#!/usr/bin/env python
"""Target of the rocprofv3 --pmc passes over t2v_wgrad_tn_group: the q|k|v and conv3x3 groups of the 320-channel level, eager, 5 launches each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2v_turbo_amd import native as nt  # noqa: E402

ops = nt.HipOps()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(3)
for M, probs in ((40960, [(320, 64)] * 3 + [(192, 320)]), (40960, [(320, 64), (576, 320)])):
    ins = [(torch.randn(M, r, device=dev, generator=gen).bfloat16(), torch.randn(M, c, device=dev, generator=gen).bfloat16()) for r, c in probs]
    plist = [(a, b, torch.zeros(a.shape[1], b.shape[1], device=dev), 1.0) for a, b in ins]
    for _ in range(5):
        ops.wgrad_tn_group(plist)
    torch.cuda.synchronize()
