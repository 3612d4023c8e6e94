#!/usr/bin/env python
"""Measurements for the rows built beyond the headline path: VAE encode / decode of a 16-frame
320x512 clip, the ModelScope denoiser step on the C5 latent, the fused flat-buffer AdamW / EMA passes.
    python tools/widen_bench.py > gpurun_out/widen_bench.json"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def main():
    from t2v_turbo_amd.ms_unet3d import UNet3DConditionModel
    from t2v_turbo_amd.vae import AutoencoderKL
    from t2v_turbo_amd.dist import FlatGradSync
    from t2v_turbo_amd.optim import FlatAdamW, update_ema_flat
    os.environ.setdefault("T2V_HIP_GRAPH", "1")
    dev = torch.device("cuda", 0)
    out = {}
    dd = dict(double_z=True, z_channels=4, resolution=512, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    with torch.device(dev):
        ae = AutoencoderKL(ddconfig=dd, embed_dim=4)
    ae = ae.to(torch.bfloat16).eval()
    z = torch.randn(1, 4, 16, 40, 64, device=dev, dtype=torch.bfloat16)
    vid = torch.randn(16, 3, 320, 512, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        ms_dec = timed(lambda: ae.decode_video(z))
        ms_enc = timed(lambda: ae.encode(vid))
    out["vae_decode_16f_320x512"] = {"ms": round(ms_dec, 2), "tflop": 25.02, "tflops": round(25.02 / ms_dec * 1e3, 1)}
    out["vae_encode_16f_320x512"] = {"ms": round(ms_enc, 2), "tflop": 11.04, "tflops": round(11.04 / ms_enc * 1e3, 1)}
    # reward-gradient branch: decode 6 frames with grad w.r.t. the latents, backward from the image gradient
    ae.requires_grad_(False)
    z6 = torch.randn(6, 4, 40, 64, device=dev, dtype=torch.bfloat16, requires_grad=True)
    gimg = torch.randn(6, 3, 320, 512, device=dev, dtype=torch.bfloat16)

    def fwd_bwd():
        z6.grad = None
        ae.decode(z6).backward(gimg)

    ms_fb = timed(fwd_bwd, iters=5, warm=2)
    ae.native_mode = "off"
    ms_fb_torch = timed(fwd_bwd, iters=3, warm=1)
    ae.native_mode = "auto"
    out["vae_decode_fwd_bwd_6f_320x512"] = {"ms_native": round(ms_fb, 2), "ms_torch_autograd_same_gpu": round(ms_fb_torch, 2),
                                            "tflop_fwd_plus_dx": round(2 * 6 * 1.5635, 2),
                                            "tflops_native": round(2 * 6 * 1.5635 / ms_fb * 1e3, 1)}
    del ae
    with torch.device(dev):
        ms_model = UNet3DConditionModel(time_cond_proj_dim=256)
    for k, v in ms_model.state_dict().items():
        if float(v.abs().max()) == 0:
            v.normal_(0.0, 0.02)
    ms_model = ms_model.to(torch.bfloat16).eval()
    x = torch.randn(1, 4, 16, 32, 32, device=dev, dtype=torch.bfloat16)
    ctx = torch.randn(1, 77, 1024, device=dev, dtype=torch.bfloat16)
    tc = torch.randn(1, 256, device=dev, dtype=torch.bfloat16)
    ts = torch.tensor([999], device=dev)
    with torch.no_grad():
        ms_step = timed(lambda: ms_model(x, ts, ctx, timestep_cond=tc), iters=10, warm=3)
    out["modelscope_unet_step_16f_256x256"] = {"ms": round(ms_step, 2), "latent": [1, 4, 16, 32, 32], "hip_graph": True}
    del ms_model
    # BASELINE config C4: T2V-Turbo-v2 sampling, 16 steps on the 200-step grid with the motion-guidance embedding (unet_mg:
    # motion_cond_proj_dim = 256, switched off below the percentage threshold: pipeline/t2v_turbo_vc2_pipeline.py:190-204) +
    # 16-frame decode; the motion-prior preprocessing that feeds it (DDIM inversion = 200 UNet forwards) is timed per forward
    import bench
    from t2v_turbo_amd.pipeline import T2VTurboVC2Pipeline, make_synthetic_t2v
    from t2v_turbo_amd.unet3d import UNetModel
    cfg = dict(bench.VC2_UNET, motion_cond_proj_dim=256)
    torch.manual_seed(1234)
    with torch.device(dev):
        unet = UNetModel(**cfg)
    g = torch.Generator(device=dev).manual_seed(1234)
    with torch.no_grad():
        for p_ in unet.parameters():
            if float(p_.abs().max()) == 0.0:
                p_.normal_(0.0, 0.02, generator=g)
    unet = unet.to(torch.bfloat16).eval()
    unet.dtype = torch.bfloat16
    pipe = T2VTurboVC2Pipeline(make_synthetic_t2v(unet, dev, torch.bfloat16), None, {"params": {"unet_config": {"params": cfg}}})
    pe = torch.randn(1, 77, 1024, device=dev, dtype=torch.bfloat16)

    def clip16():
        gen = torch.Generator(device=dev).manual_seed(42)
        return pipe(prompt=None, height=320, width=512, frames=16, fps=16, guidance_scale=7.5, motion_gs=0.1, use_motion_cond=True,
                    percentage=0.3, num_inference_steps=16, lcm_origin_steps=200, prompt_embeds=pe, generator=gen, output_type="pt")

    with torch.no_grad():
        ms_c4 = timed(clip16, iters=3, warm=2)
        vid16 = clip16()
    out["clip_16step_v2_motion_cond_16f_320x512"] = {"ms": round(ms_c4, 1), "finite": bool(torch.isfinite(vid16.float()).all()),
                                                     "video_shape": list(vid16.shape), "hip_graph": True}
    del pipe, unet
    n = 117_150_000  # LoRA parameter count of the v1 run (468.6 MB fp32)
    p = [torch.nn.Parameter(torch.randn(n, device=dev))]
    sync = FlatGradSync(p)
    opt = FlatAdamW(p, sync, lr=1e-4)
    sync.flat.normal_()
    ms_opt = timed(lambda: opt.step(max_grad_norm=1.0), iters=10)
    out["flat_adamw_clip_step_117M"] = {"ms": round(ms_opt, 3), "gb_per_s": round((n * 4 * 8) / ms_opt / 1e6, 1),
                                         "bytes_model": "sumsq read g + adamw read p,g,m,v write p,m,v = 8 x 4 B per parameter"}
    ema = torch.randn(n, device=dev)
    ms_ema = timed(lambda: update_ema_flat(ema, opt.flat_param, 0.95), iters=10)
    out["ema_update_117M"] = {"ms": round(ms_ema, 3), "gb_per_s": round(n * 4 * 3 / ms_ema / 1e6, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
