#!/usr/bin/env python
"""Device time of t2v_wgrad_tn_group at the shapes of the student's LoRA groups (MI355X), inside a hipGraph (no host overhead), with
the bytes a launch must read at least (every operand once) and the rate that corresponds to; correctness against fp32 torch on the
first group.

    python tools/wgrad_time.py [lib.so ...]        (default: the product library; several libraries are timed side by side)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2v_turbo_amd import native as nt  # noqa: E402

# name, tokens, [(R = columns of a, C = columns of b)]: dU of each leaf (dy^T t) and dD (G^T x)
GROUPS = [
    ("L0 q|k|v", 40960, [(320, 64)] * 3 + [(192, 320)]),
    ("L0 out", 40960, [(320, 64), (64, 320)]),
    ("L0 ff.proj", 40960, [(2560, 64), (64, 320)]),
    ("L0 ff.out", 40960, [(320, 64), (64, 1280)]),
    ("L0 conv3x3", 40960, [(320, 64), (576, 320)]),
    ("L0 tconv", 40960, [(320, 64), (192, 320)]),
    ("L1 q|k|v", 10240, [(640, 64)] * 3 + [(192, 640)]),
    ("L1 ff.proj", 10240, [(5120, 64), (64, 640)]),
    ("L1 conv3x3", 10240, [(640, 64), (576, 640)]),
    ("L2 q|k|v", 2560, [(1280, 64)] * 3 + [(192, 1280)]),
    ("L2 conv3x3", 2560, [(1280, 64), (576, 1280)]),
    ("L3 conv3x3", 640, [(1280, 64), (576, 1280)]),
]


def graph_us(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def ops_for(path):
    """An op backend on the library at ``path`` (the product library, or a variant build next to it)."""
    import ctypes as C
    ops = nt.HipOps()
    if os.path.abspath(path) != os.path.abspath(nt.LIB_PATH):
        lib = C.CDLL(path)
        for name, (res, args) in nt._SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        ops.lib = lib
    return ops


def main():
    libs = sys.argv[1:] or [nt.LIB_PATH]
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(3)
    print("group,M,problems,min_mbyte," + ",".join(f"us[{os.path.basename(p)}],tb_per_s" for p in libs) + ",max_rel_err")
    opss = [ops_for(p) for p in libs]
    for name, M, probs in GROUPS:
        ops_in = [(torch.randn(M, r, device=dev, generator=gen).bfloat16(), torch.randn(M, c, device=dev, generator=gen).bfloat16()) for r, c in probs]
        byts = sum(a.numel() + b.numel() for a, b in ops_in) * 2
        cells, err = [], 0.0
        for ops in opss:
            outs = [torch.zeros(a.shape[1], b.shape[1], device=dev) for a, b in ops_in]
            plist = [(a, b, o, 0.5) for (a, b), o in zip(ops_in, outs)]
            us = graph_us(lambda: ops.wgrad_tn_group(plist))
            cells += [f"{us:.1f}", f"{byts / us / 1e6:.2f}"]
            for (a, b), o in zip(ops_in, outs):
                ref = 0.5 * (a.float().t() @ b.float())
                err = max(err, float((o - ref).norm() / ref.norm()))
        print(f"{name},{M},{'+'.join(f'{r}x{c}' for r, c in probs)},{byts / 1e6:.1f}," + ",".join(cells) + f",{err:.2e}")


if __name__ == "__main__":
    main()
