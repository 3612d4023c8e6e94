#!/usr/bin/env python
"""Launch census of the native student step at FULL size (VideoCrafter2 widths, latent 1x4x16x40x64, LoRA rank 64) without a
GPU: the gradient engine is recorded against a backend whose ops do nothing but note their operand shapes.  Output: one CSV row per
distinct launch (op, shape, count, algorithmic FLOPs / bytes) and a summary — launches, FLOPs per phase, activation pool, arenas —
the planning data for tuning the step on hardware (which GEMM shapes exist, where the bytes go).

    python tools/student_step_shapes.py [--out profiles/r01_student_step_shapes.csv] [--train-mode 1]"""
import argparse
import collections
import csv
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class CensusOps:
    is_native = False
    act_dtype = torch.bfloat16

    def __init__(self):
        self.rows = collections.Counter()
        self.phase = "fwd"

    def init(self):
        pass

    def gn_ws_floats(self, *a, **k):
        return 8

    gn_bwd_ws_floats = group_norm_ws_floats = gn_ws_floats

    def gemm(self, a0, w, out, *, M, N, a1=None, mode=0, n_img=0, h=0, wd=0, frames=0, bias=None, rowvec=None, rowvec_div=0,
             residual=None, act=0, alpha=1.0, batch=1, batch_inner=1, a_strides=(0, 0), w_strides=(0, 0), o_strides=(0, 0),
             tile_cfg=0, split_k=0):
        taps = {0: 1, 4: 3}.get(mode, 9)
        K = taps * (a0.shape[1] + (0 if a1 is None else a1.shape[1]))
        flops = 2.0 * M * N * K * batch
        nbytes = batch * (M * K // taps * 2 + N * K * 2 + M * N * out.element_size() + (M * N * 2 if residual is not None else 0))
        self.rows[(self.phase, "gemm", f"mode{mode} M{M} N{N} K{K} b{batch} split{split_k} {'f32' if out.dtype == torch.float32 else 'bf16'}",
                   flops, nbytes)] += 1

    def __getattr__(self, name):
        def op(*a, **k):
            nbytes = sum(t.numel() * t.element_size() for t in a if isinstance(t, torch.Tensor))
            shape = "x".join(str(tuple(t.shape)) for t in a if isinstance(t, torch.Tensor))[:80]
            self.rows[(self.phase, name, shape, 0.0, nbytes)] += 1
        return op


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r01_student_step_shapes.csv"))
    ap.add_argument("--train-mode", type=int, default=1)
    ap.add_argument("--tiny", type=int, default=0)
    a = ap.parse_args()
    from t2v_turbo_amd import lora
    from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
    from t2v_turbo_amd.unet3d import UNetModel
    from tests.util import VC2_UNET, tiny_unet_params
    t0 = time.time()
    cfg, shape, cdim = (tiny_unet_params(), (1, 4, 4, 16, 16), 128) if a.tiny else (dict(VC2_UNET), (1, 4, 16, 40, 64), 1024)
    m = UNetModel(**cfg)
    m.requires_grad_(False)
    lora.inject_trainable_lora_extended(m, r=64)
    params = lora.lora_parameters(m)
    m.train(bool(a.train_mode))
    ops = CensusOps()
    eng = UNetGradEngine(m, ops)
    eng.bind_lora(params)
    x = torch.zeros(shape)
    ts = torch.tensor([999])
    ctx = torch.zeros(1, 77, cdim)
    tc = torch.zeros(1, 256)
    with torch.no_grad():
        emb_all = m.conditioning_emb_all(ts, 16, tc)
    print(f"model + LoRA built ({time.time() - t0:.0f}s); recording ...", flush=True)
    eng.forward_tape(x, ts, ctx, 16, tc, None, emb_all=emb_all, seed=1)
    ops.phase = "bwd"
    eng.backward(torch.zeros_like(x), flat_grad=torch.zeros(eng.lora_numel))
    plan = next(iter(eng.plans.values()))
    rows = sorted(ops.rows.items(), key=lambda kv: (-kv[0][3] * kv[1], -kv[0][4] * kv[1]))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["phase", "op", "shape", "count", "gflop_each", "mbytes_each", "gflop_total", "mbytes_total"])
        for (phase, op, shp, fl, nb), n in rows:
            w.writerow([phase, op, shp, n, round(fl / 1e9, 3), round(nb / 1e6, 3), round(fl * n / 1e9, 2), round(nb * n / 1e6, 2)])
    tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for (phase, op, shp, fl, nb), n in ops.rows.items():
        t = tot[(phase, op)]
        t[0] += n
        t[1] += fl * n
        t[2] += nb * n
    print(f"recorded in {time.time() - t0:.0f}s")
    print(f"activation pool {plan['pool_bytes'] / 2 ** 30:.2f} GiB (bf16), LoRA operand arena {eng.lp_used * 2 / 2 ** 20:.0f} MiB, "
          f"gradient arena {eng.e_used * 4 / 2 ** 20:.0f} MiB, {eng.lora_numel / 1e6:.1f} M LoRA elements, {len(eng.drop_sites)} dropout sites")
    for phase in ("fwd", "bwd"):
        n = sum(v[0] for k, v in tot.items() if k[0] == phase)
        fl = sum(v[1] for k, v in tot.items() if k[0] == phase)
        print(f"{phase}: {n} launches, {fl / 1e12:.2f} TFLOP in GEMMs")
        for (ph, op), v in sorted(tot.items(), key=lambda kv: -kv[1][2]):
            if ph == phase and v[0]:
                print(f"    {op:18s} {v[0]:5d} launches  {v[1] / 1e12:7.3f} TFLOP  {v[2] / 1e9:8.2f} GB operand bytes")


if __name__ == "__main__":
    main()
