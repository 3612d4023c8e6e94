#!/usr/bin/env python
"""Per-shape device time of every t2v_gemm launch of the full-size LoRA student's recorded forward and backward lists (MI355X):

    python tools/student_gemm_profile.py > gpurun_out/student_gemm_shapes.csv

Each recorded launch is re-issued on its own operands (the plan's buffers), 8 times back to back between two events; launches are
grouped by (list, mode, M, N, K, residual, dropout epilogue, tile, split): count, microseconds per launch, total ms per step, achieved
TFLOP/s and the HBM floor of the shape's algorithmic bytes.  This is where the un-merged LoRA forward's 23 ms of extra GEMM time
(1 036 launches for +8 % FLOPs) is itemised."""
import collections
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from t2v_turbo_amd import lora, native as nt
    from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
    from t2v_turbo_amd.native import HipOps
    from t2v_turbo_amd.unet3d import UNetModel
    dev = torch.device("cuda", 0)
    with torch.device(dev):
        m = UNetModel(**dict(bench.VC2_UNET))
    g = torch.Generator(device=dev).manual_seed(1234)
    with torch.no_grad():
        for p in m.parameters():
            if float(p.abs().max()) == 0.0:
                p.normal_(0.0, 0.02, generator=g)
    m.requires_grad_(False)
    lora.inject_trainable_lora_extended(m, r=64)
    params = lora.lora_parameters(m)
    with torch.no_grad():
        for p in params:
            if float(p.abs().max()) == 0.0:
                p.normal_(0.0, 0.01, generator=g)
    m.train()
    m.native_mode = "off"
    ops = HipOps()
    eng = UNetGradEngine(m, ops)
    eng.bind_lora(params)
    x = torch.randn(1, 4, 16, 40, 64, device=dev, generator=g)
    ts = torch.tensor([500], device=dev)
    ctx = torch.randn(1, 77, 1024, device=dev, generator=g)
    tc = torch.randn(1, 256, device=dev, generator=g)
    emb_all = m.conditioning_emb_all(ts, 16, tc).detach()
    eng.forward_tape(x, ts, ctx, 16, tc, None, emb_all=emb_all)
    flat = torch.zeros(eng.lora_numel, device=dev)
    eng.backward(torch.randn_like(x), flat_grad=flat, accumulate=False)
    torch.cuda.synchronize()
    plan = eng._last
    stream = ops.stream()
    rows = collections.OrderedDict()
    for which in ("rec", "rec_bwd"):
        for fn, args, name in plan[which]:
            if name != "t2v_gemm":
                continue
            d = args[0]._obj
            K = (d.c0 + d.c1) * {nt.GEMM_LINEAR: 1, nt.GEMM_TCONV3: 3}.get(d.mode, 9)
            key = (which, d.mode, d.M, d.N, K, d.batch, int(bool(d.residual)), int(bool(d.drop_seed)), int(bool(d.rowvec)), d.act, d.split_k, int(bool(getattr(d, "lora_t", None))))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn(*args, stream)
            e0.record()
            for _ in range(8):
                fn(*args, stream)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 8 * 1e3
            r = rows.setdefault(key, [0, 0.0])
            r[0] += 1
            r[1] += us
    print("list,mode,M,N,K,batch,res,drop,rowvec,act,split,lora_epi,count,us,total_ms,tflops,us_hbm_floor")
    out = []
    for (which, mode, M, N, K, batch, res, drop, rv, act, split, lra), (cnt, us_sum) in rows.items():
        us = us_sum / cnt
        flop = 2.0 * M * N * K * batch
        n_out = N // 2 if act == nt.ACT_GEGLU else N
        byts = 2.0 * batch * (M * K / ({nt.GEMM_LINEAR: 1, nt.GEMM_TCONV3: 3}.get(mode, 9)) + N * K + M * n_out * (1 + res))
        out.append((us_sum, f"{which},{mode},{M},{N},{K},{batch},{res},{drop},{rv},{act},{split},{lra},{cnt},{us:.2f},{us_sum / 1e3:.3f},"
                            f"{flop / us / 1e6:.1f},{byts / 8e12 * 1e6:.2f}"))
    for _, line in sorted(out, key=lambda t: -t[0]):
        print(line)
    tot = {w: sum(v[1] for k, v in rows.items() if k[0] == w) / 1e3 for w in ("rec", "rec_bwd")}
    n = {w: sum(v[0] for k, v in rows.items() if k[0] == w) for w in ("rec", "rec_bwd")}
    print(f"# forward: {n['rec']} t2v_gemm launches {tot['rec']:.2f} ms; backward: {n['rec_bwd']} launches {tot['rec_bwd']:.2f} ms "
          "(each launch timed alone, warm caches, 8 repeats)")


if __name__ == "__main__":
    main()
