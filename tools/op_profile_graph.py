#!/usr/bin/env python
"""In-graph device time of every NON-GEMM launch class of the UNet step (R back-to-back launches of one recorded
call inside a hipGraph: no host launch overhead), with algorithmic bytes and the resulting GB/s.

    python tools/op_profile_graph.py [--out gpurun_out/op_profile.csv]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gemm_profile_graph import graph_time  # noqa: E402


def algo_bytes(name, a):
    if name in ("t2v_gn_stats", "t2v_gn_apply", "t2v_group_norm"):
        return {"t2v_gn_stats": 1, "t2v_gn_apply": 2, "t2v_group_norm": 2}[name] * 2.0 * a[6] * a[7] * (a[1] + a[4])
    if name == "t2v_group_norm_cs":   # (cs0, cs1, x0, c0, ld0, x1, c1, ld1, units, rows, ...): one read + one write of the tensor
        return 2 * 2.0 * a[8] * a[9] * (a[3] + a[6])
    if name == "t2v_layernorm":
        return 4.0 * a[2] * a[3]
    if name == "t2v_attn_temporal":
        return 8.0 * a[8] * a[9] * a[10] * a[11] * 64
    if name == "t2v_attn_spatial":
        n_img, sq, skv, heads, kv_div = a[9], a[10], a[11], a[12], a[13]
        return 2.0 * 64 * heads * (2 * n_img * sq + 2 * (n_img // kv_div) * skv)
    return 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "op_profile.csv"))
    args = ap.parse_args()
    import bench

    dev = torch.device("cuda", 0)
    model = bench.build_model(dev, torch.bfloat16)
    x, ctx, tc = bench.synth_inputs(dev, torch.bfloat16)
    with torch.no_grad():
        model(x, torch.tensor([999], device=dev), context=ctx, fps=16, timestep_cond=tc)
    rec = next(iter(model.native_engine().plans.values()))["rec"]
    seen = {}
    for fn, a, name in rec:
        if name == "t2v_gemm":
            continue
        key = (name,) + tuple(v for v in a if isinstance(v, (int, float)) and not isinstance(v, bool) and abs(v) < 2 ** 31)
        if key in seen:
            seen[key][0] += 1
        else:
            seen[key] = [1, fn, a]
    rows = []
    for key, (count, fn, a) in seen.items():
        us = graph_time(lambda: fn(*a, torch.cuda.current_stream().cuda_stream))
        b = algo_bytes(key[0], a)
        rows.append(dict(name=key[0], args=" ".join(str(v) for v in key[1:]), count=count, us=round(us, 2),
                         total_ms=round(us * count / 1e3, 3), mbyte=round(b / 1e6, 2),
                         gb_per_s=round(b / us / 1e3, 1) if b else ""))
    rows.sort(key=lambda r: -r["total_ms"])
    by = {}
    for r in rows:
        t = by.setdefault(r["name"], [0, 0.0, 0.0])
        t[0] += r["count"]; t[1] += r["total_ms"]; t[2] += r["mbyte"] * r["count"]
    with open(args.out, "w") as f:
        for n, (c, ms, mb) in sorted(by.items(), key=lambda kv: -kv[1][1]):
            line = f"# {n}: {c} launches, {ms:.3f} ms in-graph, {mb / 1e3:.3f} GB -> {mb / ms / 1e3 if ms else 0:.2f} TB/s"
            print(line)
            f.write(line + "\n")
        cols = list(rows[0].keys())
        f.write(",".join(cols) + "\n")
        for r in rows:
            f.write(",".join(str(r[c]) for c in cols) + "\n")


if __name__ == "__main__":
    main()
