#!/usr/bin/env python
"""Time (and check) the full-size KL-VAE decode by itself: 16 latent frames (1,4,16,40,64) -> (1,3,16,320,512), HIP events around
replays of the recorded plan, one-frame parity against the fp32 oracle, and a per-kernel census of the plan (which convs went to
t2v_conv_halo).  Much cheaper than bench.py's clip leg when the question is only the decoder (A/B by environment, e.g.
T2V_VAE_HALO=0 / 1).

    python tools/vae_time.py [--reps 5] [--parity 1]
"""
import argparse
import collections
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--parity", type=int, default=1)
    args = ap.parse_args()
    from t2v_turbo_amd.vae import AutoencoderKL
    dev, dtype = torch.device("cuda", 0), torch.bfloat16
    dd = dict(double_z=True, z_channels=4, resolution=512, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
              attn_resolutions=[], dropout=0.0)
    torch.manual_seed(7)
    with torch.device(dev):
        vae = AutoencoderKL(ddconfig=dd, embed_dim=4)
    vae = vae.to(dtype).eval()
    gz = torch.Generator().manual_seed(5)
    out = {"env": {k: os.environ[k] for k in ("T2V_VAE_HALO", "T2V_CONV_HALO", "T2V_HIP_LIB") if k in os.environ}}
    if args.parity:
        from oracle import vae_oracle as vo
        z = torch.randn(1, 4, 1, 40, 64, generator=gz) * 0.18215 * 4.0
        sd = {k: v.detach().float().cpu() for k, v in vae.state_dict().items()}
        ref = vo.decode_first_stage_2dae(sd, dd, z)
        with torch.no_grad():
            v = vae.decode_video(z.to(dev, dtype))
        out["parity_rel_l2"] = float((v.float().cpu() - ref).double().norm() / ref.double().norm())
    zs = torch.randn(1, 4, 16, 40, 64, generator=gz).to(dev, dtype) * 0.18215 * 4.0
    with torch.no_grad():
        vae.decode_video(zs)
        vae.decode_video(zs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            vae.decode_video(zs)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    out.update(ms=round(ms, 3), tflops=round(25.02 / ms * 1e3, 1), frac=round(25.02 / ms * 1e3 / 2500.0, 4))
    eng = vae._engine_box.engine
    if eng is not None:
        plan = [p for p in eng.plans.values() if "rec" in p][-1]
        out["launches"] = dict(collections.Counter(e[2] for e in plan["rec"]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
