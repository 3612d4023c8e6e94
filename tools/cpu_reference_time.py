#!/usr/bin/env python
"""CPU baseline as SURVEY.md 8(d) / BASELINE.md 3 specify it: the IMPORTED reference ``UNetModel`` (fp32, eval,
use_checkpoint=False, vanilla attention, no_grad) on the bench's synthetic inputs, 1 warm-up + 3 runs, median — next to the
oracle (``oracle.unet_oracle``, the CPU restatement bench.py times on the GPU box, where /root/reference does not exist) on
the same weights, inputs and threads.  Runs where the reference checkout is (this container), writes one JSON file.

    python tools/cpu_reference_time.py --frames 16 --out profiles/r02_cpu_reference_timing.json
"""
import argparse
import json
import os
import platform
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--decode-frames", type=int, default=0, help="also time the KL-VAE decode of this many 320x512 frames (reported per 16 frames)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_cpu_reference_timing.json"))
    a = ap.parse_args()
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden as mg
    mg.install_stubs()
    from lvdm.modules.networks.openaimodel3d import UNetModel as RefUNet
    import bench
    from oracle import unet_oracle as uo
    from t2v_turbo_amd.nn_util import guidance_embedding

    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    cfg = mg.full_unet_params()
    assert {k: cfg[k] for k in bench.VC2_UNET if k in cfg} == {k: (list(v) if isinstance(v, tuple) else v)
                                                              for k, v in bench.VC2_UNET.items() if k in cfg}, "bench config differs from the yaml"
    torch.manual_seed(1234)
    ref = RefUNet(**cfg).eval()
    g = torch.Generator().manual_seed(1234)
    with torch.no_grad():
        for p in ref.parameters():  # zero_module'd tensors re-drawn, as bench.build_model does
            if float(p.abs().max()) == 0.0:
                p.normal_(0.0, 0.02, generator=g)
    sd = {k: v.detach().float() for k, v in ref.state_dict().items()}
    gi = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 16, 40, 64, generator=gi)[:, :, :a.frames].contiguous()
    ctx = torch.randn(1, 77, 1024, generator=gi)
    tc = guidance_embedding(torch.tensor([7.5]), 256)
    ts = torch.tensor([999])

    def run_ref():
        with torch.no_grad():
            return ref(x, ts, context=ctx, fps=16, timestep_cond=tc)

    def run_oracle():
        return uo.unet_forward(sd, bench.VC2_UNET, x, ts, ctx, fps=16, timestep_cond=tc)

    out = {"frames": a.frames, "threads": threads, "cpu": platform.processor() or platform.machine(), "runs": a.runs}
    try:
        with open("/proc/cpuinfo") as f:
            out["cpu"] = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:  # noqa: BLE001
        pass
    res = {}
    for name, fn in (("reference", run_ref), ("oracle", run_oracle)):
        t0 = time.time()
        y = fn()
        warm = time.time() - t0
        times = []
        for _ in range(a.runs):
            t0 = time.time()
            y = fn()
            times.append(time.time() - t0)
        res[name] = y
        out[name] = {"warmup_s": round(warm, 2), "runs_s": [round(t, 2) for t in times], "median_s": round(statistics.median(times), 2),
                     "steps_per_s_16f_equiv": round((a.frames / 16.0) / statistics.median(times), 5)}
        print(name, out[name], flush=True)
    d = (res["oracle"] - res["reference"]).double().norm() / res["reference"].double().norm()
    out["oracle_vs_reference_rel_l2"] = float(d)
    out["oracle_over_reference_time"] = round(out["oracle"]["median_s"] / out["reference"]["median_s"], 3)
    if a.decode_frames:
        # the reference's KL-VAE decode (frame loop of LatentDiffusion.decode_first_stage_2DAE, ddpm3d.py:666-679)
        # and the 4-step clip total = 4 UNet forwards + the 16-frame decode, reference vs oracle
        import yaml
        from lvdm.models.autoencoder import AutoencoderKL as RefAE
        from oracle import vae_oracle as vo
        cfgfull = yaml.safe_load(open(os.path.join(mg.REF, "configs/inference_t2v_512_v2.0.yaml")))
        fs = cfgfull["model"]["params"]["first_stage_config"]["params"]
        torch.manual_seed(4321)
        ae = RefAE(**fs).eval()
        ae_sd = {k: v.detach().float() for k, v in ae.state_dict().items()}
        f = a.decode_frames
        z = torch.randn(1, 4, f, 40, 64, generator=torch.Generator().manual_seed(3)) * 0.18215

        def dec_ref():
            with torch.no_grad():
                zz = z / 0.18215
                return torch.cat([ae.decode(zz[:, :, i]).unsqueeze(2) for i in range(f)], dim=2)

        def dec_oracle():
            return vo.decode_first_stage_2dae(ae_sd, fs["ddconfig"], z)

        dres = {}
        for name, fn in (("reference", dec_ref), ("oracle", dec_oracle)):
            fn()
            times = []
            for _ in range(a.runs):
                t0 = time.time()
                dres[name] = fn()
                times.append(time.time() - t0)
            med = statistics.median(times)
            out["decode_" + name] = {"frames": f, "runs_s": [round(t, 2) for t in times], "median_s_per_frame": round(med / f, 2),
                                     "s_per_16_frames": round(med / f * 16, 1)}
            print("decode", name, out["decode_" + name], flush=True)
        out["decode_oracle_vs_reference_rel_l2"] = float((dres["oracle"] - dres["reference"]).double().norm() / dres["reference"].double().norm())
        for name in ("reference", "oracle"):
            out["clip_4step_s_" + name] = round(4 * out[name]["median_s"] * 16.0 / a.frames + out["decode_" + name]["s_per_16_frames"], 1)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
