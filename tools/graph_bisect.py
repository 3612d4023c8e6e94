#!/usr/bin/env python
"""Debug tool: find the first launch of the training engine's forward (or backward) list whose result differs when the
list's prefix up to and including it is replayed from a captured hipGraph instead of launch by launch.

    python tools/graph_bisect.py [--tiny 1] [--which rec|rec_bwd]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiny", type=int, default=0)
    ap.add_argument("--which", default="rec")
    ap.add_argument("--train", type=int, default=1, help="train mode (dropouts active)")
    ap.add_argument("--replays", type=int, default=2, help="replays of the captured prefix before the comparison")
    a = ap.parse_args()
    import bench
    from t2v_turbo_amd import lora
    from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
    from t2v_turbo_amd.native import HipOps
    from t2v_turbo_amd.unet3d import UNetModel

    dev = torch.device("cuda", 0)
    cfg = dict(bench.VC2_UNET)
    shape, ctx_dim = (1, 4, 16, 40, 64), 1024
    if a.tiny:
        cfg.update(model_channels=64, context_dim=128)
        shape, ctx_dim = (1, 4, 4, 16, 16), 128
    with torch.device(dev):
        student = UNetModel(**cfg)
    g = torch.Generator(device=dev).manual_seed(1234)
    with torch.no_grad():
        for p in student.parameters():
            if float(p.abs().max()) == 0.0:
                p.normal_(0.0, 0.02, generator=g)
    student.requires_grad_(False)
    lora.inject_trainable_lora_extended(student, r=64)
    params = lora.lora_parameters(student)
    with torch.no_grad():
        for p in params:
            if float(p.abs().max()) == 0.0:
                p.normal_(0.0, 0.01, generator=g)
    student.train(bool(a.train))
    student.native_mode = "off"
    ops = HipOps()
    eng = UNetGradEngine(student, ops)
    eng.flash_attn_bwd = eng.tn_wgrad = True
    eng.bind_lora(params)
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(shape, generator=gen).to(dev)
    ctx = torch.randn(1, 77, ctx_dim, generator=gen).to(dev)
    ts = torch.tensor([499], device=dev)
    from t2v_turbo_amd.nn_util import guidance_embedding
    tc = guidance_embedding(torch.tensor([7.5]), 256).to(dev)
    with torch.no_grad():
        emb_all = student.conditioning_emb_all(ts, 16, tc)
    flat = torch.zeros(eng.lora_numel, device=dev)

    def fwd():
        return eng.forward_tape(x, ts, ctx, 16, tc, None, emb_all=emb_all, seed=1234)

    y0 = fwd()          # records
    y1 = fwd()          # plain replay
    torch.cuda.synchronize()
    print("record vs plain replay bit-identical:", torch.equal(y0, y1), "finite:", bool(torch.isfinite(y1).all()), flush=True)
    dout = torch.randn_like(y1)
    dx0 = eng.backward(dout, flat_grad=flat, accumulate=False)
    fwd()
    dx1 = eng.backward(dout, flat_grad=flat, accumulate=False)
    torch.cuda.synchronize()
    print("backward record vs replay bit-identical:", torch.equal(dx0, dx1), "finite:", bool(torch.isfinite(dx1).all()), flush=True)
    plan = eng._last
    rec = plan[a.which]
    out_key = "out" if a.which == "rec" else "dx"
    L = len(rec)
    print(f"{a.which}: {L} launches", flush=True)

    def prepare():
        if a.which == "rec_bwd":   # the backward consumes the forward's tape: re-run the forward first
            ops.replay(plan["rec"], ops.stream())
            plan["static"]["dout"].copy_(dout)

    prepare()
    ops.replay(rec, ops.stream())
    torch.cuda.synchronize()
    ref = plan[out_key].clone()
    prepare()
    ops.replay(rec, ops.stream())
    torch.cuda.synchronize()
    print("plain replay twice bit-identical:", torch.equal(ref, plan[out_key]), flush=True)

    def run(n):
        prepare()
        torch.cuda.synchronize()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph):
            ops.replay(rec[:n], ops.stream())
        for _ in range(a.replays):   # the engine replays ONE graph instance step after step
            prepare()
            gph.replay()
            ops.replay(rec[n:], ops.stream())
            torch.cuda.synchronize()
        out = plan[out_key]
        same = torch.equal(out, ref)
        if not same:
            fin = bool(torch.isfinite(out).all())
            err = float((out.float() - ref.float()).norm() / ref.float().norm()) if fin else float("nan")
            print(f"  n={n}: differs (finite {fin}, rel {err:.3e})", flush=True)
        else:
            print(f"  n={n}: identical", flush=True)
        del gph
        return same

    if run(L):
        print(f"whole list in one graph, replayed {a.replays}x, reproduces the plain replay: no capture problem in this list")
        return
    lo, hi = 0, L   # run(lo) passes (nothing in the graph), run(hi) fails
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if run(mid):
            lo = mid
        else:
            hi = mid
    fn, args, name = rec[hi - 1]
    print(f"first launch that breaks under capture: index {hi - 1} of {L}: {name}")
    for k in range(max(0, hi - 4), min(L, hi + 2)):
        f2, a2, n2 = rec[k]
        desc = ""
        if n2 == "t2v_gemm":
            d = a2[0]._obj
            desc = f"mode {d.mode} M {d.M} N {d.N} c0 {d.c0} c1 {d.c1} batch {d.batch} act {d.act} cfg {d.tile_cfg} split {d.split_k} out_f32 {d.out_f32}"
        else:
            desc = " ".join(str(v) for v in a2 if isinstance(v, (int, float)))[:160]
        print(f"   [{k}] {n2} {desc}")


if __name__ == "__main__":
    main()
