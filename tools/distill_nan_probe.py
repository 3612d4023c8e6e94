#!/usr/bin/env python
"""Debug tool: run the native-student distillation step and report where the first non-finite value appears.
Phases: A plain replay without parameter updates, B plain replay with AdamW, C hipGraph with AdamW.  Per step: loss,
finiteness of the student / target predictions, of the flat LoRA gradient and of the parameters; on the first bad step the
LoRA tensors whose gradient is non-finite are listed by leaf name."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--phases", default="A,B,C")
    ap.add_argument("--flash", type=int, default=1)
    ap.add_argument("--tn", type=int, default=1)
    a = ap.parse_args()
    import bench
    from t2v_turbo_amd import cd_math, dist as tdist, lora
    from t2v_turbo_amd.distill import distill_step
    from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
    from t2v_turbo_amd.native import HipOps
    from t2v_turbo_amd.optim import FlatAdamW
    from t2v_turbo_amd.scheduler import T2VTurboScheduler
    from t2v_turbo_amd.unet3d import UNetModel

    dev = torch.device("cuda", 0)
    cfg = dict(bench.VC2_UNET)
    with torch.device(dev):
        student = UNetModel(**dict(cfg, use_checkpoint=True))
        teacher = UNetModel(**dict(cfg, time_cond_proj_dim=None))
    g = torch.Generator(device=dev).manual_seed(1234)
    with torch.no_grad():
        for m in (student, teacher):
            for p in m.parameters():
                if float(p.abs().max()) == 0.0:
                    p.normal_(0.0, 0.02, generator=g)
    teacher = teacher.to(torch.bfloat16).eval().requires_grad_(False)
    teacher.dtype = torch.bfloat16
    student.requires_grad_(False)
    lora.inject_trainable_lora_extended(student, r=64)
    params = lora.lora_parameters(student)
    names = {}
    for n, mod in student.named_modules():
        if hasattr(mod, "lora_up"):
            names[id(mod.lora_up.weight)] = n + ".lora_up"
            names[id(mod.lora_down.weight)] = n + ".lora_down"
    sync = tdist.FlatGradSync(params)
    student.train()
    student.native_mode = "off"
    with torch.no_grad():
        for p in params:
            if float(p.abs().max()) == 0.0:
                p.normal_(0.0, 0.01, generator=g)
    opt = FlatAdamW(params, sync, lr=1e-5)
    sched = T2VTurboScheduler()
    solver = cd_math.DDIMSolver(sched.alphas_cumprod.numpy(), ddim_timesteps=50).to(dev)
    gen = torch.Generator().manual_seed(0)
    lat = torch.randn((1, 4, 16, 40, 64), generator=gen).to(dev) * 0.18215
    pe, ue = torch.randn(1, 77, 1024, generator=gen).to(dev), torch.randn(1, 77, 1024, generator=gen).to(dev)

    def fin(t):
        return bool(torch.isfinite(t).all())

    for phase in a.phases.split(","):
        eng = UNetGradEngine(student, HipOps())
        eng.flash_attn_bwd, eng.tn_wgrad, eng.use_graph = bool(a.flash), bool(a.tn), phase == "C"
        eng.bind_lora(params)
        for s in range(a.steps):
            loss, info = distill_step(student, teacher, solver, sched, lat, pe, ue, optimizer=None if phase == "A" else opt,
                                      grad_sync=sync, autocast_dtype=torch.bfloat16, student_engine=eng)
            torch.cuda.synchronize()
            flat_ok, par_ok = fin(sync.flat), fin(opt.flat_param)
            print(f"phase {phase} step {s}: idx {int(info['index'][0])} t {int(info['start_timesteps'][0])} loss {float(loss):.4f} "
                  f"grad_norm {float(info['grad_norm']):.4g} grad_finite {flat_ok} params_finite {par_ok} "
                  f"dx_emb_finite {fin(eng.d_emb_all)}", flush=True)
            if not (flat_ok and par_ok):
                off, bad = 0, []
                for p in sync.params:
                    gslice = sync.flat[off:off + p.numel()]
                    if not fin(gslice):
                        bad.append((names.get(id(p), "?"), int((~torch.isfinite(gslice)).sum()), p.numel()))
                    off += p.numel()
                print(f"  non-finite gradient tensors: {len(bad)} of {len(sync.params)}; first: {bad[:12]}", flush=True)
                plan = eng._last
                print(f"  student out finite {fin(plan['out'])}, dx finite {fin(plan['dx'])}", flush=True)
                break
        eng = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        if not fin(opt.flat_param):
            print("parameters are non-finite: stop")
            break


if __name__ == "__main__":
    main()
