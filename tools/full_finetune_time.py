#!/usr/bin/env python
"""The full fine-tuning student (row a20: every UNet parameter trainable, no LoRA — train_latent_t2v_turbo_v2.py:669,798-816,1262) at the
FULL VideoCrafter2 width and the bench latent (1,4,16,40,64) on the device: forward + backward through the module route
(`unet(...)`, `loss.backward()`), an SGD-style update of every weight between steps (so that the in-place pack refresh is inside the
timed loop), peak memory, launches per list, finiteness of every gradient.  Correctness at this width rests on the kernel tests and
the tiny-width fixture test; what this adds is that the path RUNS at the size the reference trains at (2 560-channel concat GroupNorms,
10 240-column GEGLU pre-activations, 472 MB im2col matrices) and what it costs.

    python tools/full_finetune_time.py [--frames 16] [--steps 3] [--train 1]
"""
import argparse
import json
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--train", type=int, default=1, help="train mode (the TemporalConvBlock dropouts live), as the v2 script runs its student")
    ap.add_argument("--native", type=int, default=1, help="0: the torch composite path over ATen kernels (native_mode = 'off'), for comparison")
    args = ap.parse_args()
    import bench
    dev = torch.device("cuda", 0)
    m = bench.build_model(dev, torch.float32)
    m.requires_grad_(True)
    m.train() if args.train else m.eval()
    if not args.native:
        m.native_mode = "off"
    x, ctx, tc = bench.synth_inputs(dev, torch.float32)
    x = x[:, :, :args.frames].contiguous()
    ts = torch.tensor([999], device=dev)
    params = [p for p in m.parameters()]
    out = {"frames": args.frames, "train_mode": bool(args.train), "params_m": round(sum(p.numel() for p in params) / 1e6, 1)}
    times = []
    torch.cuda.reset_peak_memory_stats()
    for step in range(args.steps + 1):   # step 0 records the two launch lists
        for p in params:
            p.grad = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            if args.native:
                warnings.simplefilter("error")    # the torch-composite route warns: it must not be taken
            y = m(x, ts, context=ctx, fps=16, timestep_cond=tc)
        loss = y.float().pow(2).mean()
        loss.backward()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        with torch.no_grad():
            for p in params:
                p.add_(p.grad, alpha=-1e-6)
        torch.cuda.synchronize()
        times.append((t1 - t0) * 1e3)
        if step == 0:
            out["all_grads_present"] = all(p.grad is not None for p in params)
            out["all_grads_finite"] = all(bool(torch.isfinite(p.grad).all()) for p in params)
            out["grad_norm"] = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params)))
    out["grad_norm_last_step"] = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params)))   # (after steps weight updates)
    if args.native:   # the pack refresh alone (inside every timed step's forward): host time to issue it, and until the device is done
        eng = m._engine_box.full
        best = (1e9, 1e9)
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng._refresh() if hasattr(eng, "_refresh") else eng.pk.refresh(eng.ops)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            best = min(best, ((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
        out["pack_refresh_ms"] = {"host_issue": round(best[0], 2), "until_device_done": round(best[1], 2), "packs": len(eng.pk.makers)}
    out.update(path="native gradient engine" if args.native else "torch composite (ATen kernels)", record_ms=round(times[0], 1),
               step_ms=[round(t, 1) for t in times[1:]], loss=float(loss.detach()), peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1))
    if args.native:
        eng = m._engine_box.full
        plan = next(iter(eng.plans.values()))
        out.update(launches={"forward": len(plan["rec"]), "backward": len(plan["rec_bwd"])}, plans=len(eng.plans),
                   pool_gb=round(plan["pool_bytes"] / 2 ** 30, 1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
