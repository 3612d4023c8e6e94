#!/usr/bin/env python
"""Per-shape device time of every t2v_gemm launch of the UNet step, measured inside a hipGraph (R back-to-back
launches of the recorded descriptor, so no host launch overhead), next to a plain hipBLASLt GEMM of the same
(M, N, K) via torch.mm and the two roofline floors (MFMA 2.5 PF/s dense bf16, HBM 8 TB/s on algorithmic bytes).

    python tools/gemm_profile_graph.py [--blas 1] [--out gpurun_out/gemm_profile.csv]

Ablation sweep (where does each shape's time go): build the library with the switches compiled in and point the
loader at it, then ask for the bit sets to time next to the full kernel (bits: t2v-turbo_amd/csrc/gemm.hip, ABL):

    T2V_EXTRA_HIPCC_FLAGS=-DT2V_GEMM_ABLATE T2V_HIP_LIB_OUT=t2v-turbo_amd/libt2v_hip_ablate.so python t2v-turbo_amd/csrc/build.py --force
    T2V_HIP_LIB=t2v-turbo_amd/libt2v_hip_ablate.so python tools/gemm_profile_graph.py --blas 0 --ablate 128,4,32,64 --top 30

Other tile ids on the recorded descriptors (e.g. the experimental 24-29 against the tuned choice):

    python tools/gemm_profile_graph.py --blas 0 --force-cfgs 24,25,26,27,28,29 --top 30
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def graph_time(fn, reps=20, replays=4):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(replays):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blas", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "gemm_profile.csv"))
    ap.add_argument("--ablate", default="", help="comma list of ablation bit sets to time per shape (ablate build only)")
    ap.add_argument("--force-cfgs", default="", help="comma list of tile ids to time every shape under as well (e.g. 24,25,27)")
    ap.add_argument("--top", type=int, default=0, help="only the N shapes with the largest launch count x FLOPs (0 = all)")
    args = ap.parse_args()
    import bench
    from t2v_turbo_amd import native as nt

    dev = torch.device("cuda", 0)
    model = bench.build_model(dev, torch.bfloat16)
    x, ctx, tc = bench.synth_inputs(dev, torch.bfloat16)
    with torch.no_grad():
        model(x, torch.tensor([999], device=dev), context=ctx, fps=16, timestep_cond=tc)
    rec = next(iter(model.native_engine().plans.values()))["rec"]
    seen = {}
    for fn, a, name in rec:
        if name not in bench.GEMM_FAMILY:   # t2v_gemm, t2v_conv_halo, t2v_linear_pr: one descriptor type
            continue
        d = a[0]._obj
        taps = {nt.GEMM_LINEAR: 1, nt.GEMM_TCONV3: 3}.get(d.mode, 9)
        K = taps * (d.c0 + d.c1)
        key = (d.mode, d.M, d.N, K, max(d.batch, 1), d.act, bool(d.residual), bool(d.rowvec), name, int(d.ln_in))
        if key in seen:
            seen[key][0] += 1
        else:
            seen[key] = [1, fn, a, d]
    rows = []
    ablate = [int(b) for b in args.ablate.split(",") if b]
    lib = nt.load()
    items = list(seen.items())
    if args.top:
        items.sort(key=lambda kv: -kv[1][0] * (kv[0][1] * kv[0][2] * (kv[0][3] + 2000.0)))
        items = items[:args.top]
    for key, (count, fn, a, d) in items:
        mode, M, N, K, batch, act, has_res, has_rv, kname, ln_in = key
        s = torch.cuda.current_stream().cuda_stream
        us = graph_time(lambda: fn(*a, torch.cuda.current_stream().cuda_stream))
        n_out = N // 2 if act == nt.ACT_GEGLU else N
        flops = 2.0 * M * N * K * batch
        src_rows = M if mode in (nt.GEMM_LINEAR, nt.GEMM_CONV3X3, nt.GEMM_TCONV3) else (M * 4 if "S2" in str(mode) else M)
        if mode in (nt.GEMM_CONV3X3_S2, nt.GEMM_CONV3X3_S2_PAD01):
            src_rows = M * 4
        elif mode == nt.GEMM_CONV3X3_UP2:
            src_rows = M // 4
        cin = d.c0 + d.c1
        byts = batch * (src_rows * cin * 2 + N * K * 2 + M * n_out * 2 * (2 if has_res else 1))
        us_blas = None
        if args.blas and batch == 1:
            A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            B = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
            Cc = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            try:
                us_blas = graph_time(lambda: torch.mm(A, B.t(), out=Cc), reps=10, replays=3)
            except Exception as e:  # noqa
                us_blas = None
            del A, B, Cc
        rows.append(dict(kernel=kname.replace("t2v_", ""), ln_in=ln_in, mode=mode, M=M, N=N, K=K, batch=batch, act=act, res=int(has_res), rv=int(has_rv), count=count,
                         cfg=d.tile_cfg, split=d.split_k,
                         us=round(us, 2), total_ms=round(us * count / 1e3, 3), tflops=round(flops / us / 1e6, 1),
                         us_blas=None if us_blas is None else round(us_blas, 2),
                         us_mfma_floor=round(flops / 2.5e15 * 1e6, 2), us_hbm_floor=round(byts / 8e12 * 1e6, 2)))
        for bits in ablate:  # same descriptor, parts of the kernel switched off
            lib.t2v_gemm_debug(bits)
            rows[-1][f"us_abl{bits}"] = round(graph_time(lambda: fn(*a, torch.cuda.current_stream().cuda_stream)), 2)
        lib.t2v_gemm_debug(0)
        for c in [int(v) for v in args.force_cfgs.split(",") if v]:  # same descriptor under another tile id (split-K as tuned)
            lib.t2v_gemm_force_config(c)
            try:
                ok = fn(*a, torch.cuda.current_stream().cuda_stream) == 0
                rows[-1][f"us_cfg{c}"] = round(graph_time(lambda: fn(*a, torch.cuda.current_stream().cuda_stream)), 2) if ok else None
            finally:
                lib.t2v_gemm_force_config(0)
        print(rows[-1], flush=True)
    rows.sort(key=lambda r: -r["total_ms"])
    tot = sum(r["total_ms"] for r in rows)
    floor = sum(max(r["us_mfma_floor"], r["us_hbm_floor"]) * r["count"] for r in rows) / 1e3
    blas = sum((r["us_blas"] or r["us"]) * r["count"] for r in rows) / 1e3
    blas_txt = f"; plain hipBLASLt GEMMs of the same MNK {blas:.2f} ms" if args.blas else ""
    lin = sum(r["total_ms"] for r in rows if r["mode"] == nt.GEMM_LINEAR)
    by_k = {k: round(sum(r["total_ms"] for r in rows if r["kernel"] == k), 3) for k in sorted({r["kernel"] for r in rows})}
    tf = sum(2.0 * r["M"] * r["N"] * r["K"] * r["batch"] * r["count"] for r in rows) / 1e12
    blas_txt += f"; LINEAR-mode launches {lin:.2f} ms; by kernel {by_k}; {tf:.2f} TFLOP -> {tf / tot * 1e3:.0f} TFLOP/s = {tf / tot * 1e3 / 2500:.3f} of 2.5 PF"
    print(f"GEMM per UNet step: {tot:.2f} ms in-graph; roofline floor {floor:.2f} ms{blas_txt}")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write(f"# GEMM per UNet step: {tot:.2f} ms in-graph; roofline floor {floor:.2f} ms{blas_txt}\n")
        cols = list(rows[0].keys())
        f.write(",".join(cols) + "\n")
        for r in rows:
            f.write(",".join("" if r[c] is None else str(r[c]) for c in cols) + "\n")


if __name__ == "__main__":
    main()
