#!/usr/bin/env python
"""Time the v1 consistency-distillation step (BASELINE config C3) on synthetic 16x40x64 latents.

    python tools/distill_bench.py --steps 3                       # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/distill_bench.py

Per rank: B=1; student = UNet + LoRA r=64 (torch autograd path, activation checkpointing, bf16 autocast),
teacher = frozen bf16 UNet on the native HIP engine (2 of the 4 forwards), gradients in ONE flat buffer
all-reduced over RCCL, AdamW on the LoRA tensors.  Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tiny", type=int, default=0, help="toy widths (plumbing check)")
    ap.add_argument("--rank-r", type=int, default=64)
    ap.add_argument("--native-student", type=int, default=0,
                    help="1: student forward / target forward / backward on the native gradient engine (train mode, native dropout)")
    ap.add_argument("--module-route", type=int, default=0,
                    help="1: no explicit engine — the student is called as a module (native_mode auto) and autograd drives the backward")
    ap.add_argument("--batch-teacher", type=int, default=0, help="1: teacher cond + uncond forwards as one 2-clip call")
    ap.add_argument("--native-variants", default="",
                    help="comma list of engine variants timed one after the other on ONE model build, e.g. "
                         "'plain,graph,flash,flash+tn,flash+tn+graph' (flash = T2V_FLASH_ATTN_BWD, tn = T2V_TN_WGRAD, graph = hipGraph "
                         "capture of the launch lists); implies --native-student 1")
    a = ap.parse_args()
    if a.native_variants:
        a.native_student = 1
    import bench
    from t2v_turbo_amd import cd_math, dist as tdist, lora
    from t2v_turbo_amd.distill import distill_step
    from t2v_turbo_amd.scheduler import T2VTurboScheduler
    from t2v_turbo_amd.unet3d import UNetModel

    tdist.init_distributed()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = dict(bench.VC2_UNET)
    shape, ctx_dim = (1, 4, 16, 40, 64), 1024
    if a.tiny:
        cfg.update(model_channels=64, context_dim=128)
        shape, ctx_dim = (1, 4, 4, 16, 16), 128
    t0 = time.time()
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
    lora.inject_trainable_lora_extended(student, r=a.rank_r)
    params = lora.lora_parameters(student)
    sync = tdist.FlatGradSync(params)
    eng = None
    if a.native_student:
        from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
        from t2v_turbo_amd.native import HipOps
        from t2v_turbo_amd.optim import FlatAdamW
        student.train()                      # LoRA / temporal-conv dropouts: counter-based masks on the engine
        student.native_mode = "off"
        with torch.no_grad():                # lora_up starts at zero; give the down gradients something to do
            for p in params:
                if float(p.abs().max()) == 0.0:
                    p.normal_(0.0, 0.01, generator=g)
        opt = FlatAdamW(params, sync, lr=1e-5)   # parameters / gradients / moments: three flat buffers, one fused kernel
        eng = UNetGradEngine(student, HipOps())
        eng.bind_lora(params)
    elif a.module_route:
        # what train_t2v_turbo_v1_lora.py does: plain module calls + loss.backward(); UNetModel.forward routes the LoRA student to
        # the native gradient engine by itself (native_mode = "auto")
        from t2v_turbo_amd.optim import FlatAdamW
        student.train()
        with torch.no_grad():
            for p in params:
                if float(p.abs().max()) == 0.0:
                    p.normal_(0.0, 0.01, generator=g)
        opt = FlatAdamW(params, sync, lr=1e-5)
    else:
        student.train()
        student.native_mode = "off"  # train-mode dropout + autograd: torch path
        opt = torch.optim.AdamW(params, lr=1e-5)
    sched = T2VTurboScheduler()
    solver = cd_math.DDIMSolver(sched.alphas_cumprod.numpy(), ddim_timesteps=50).to(dev)
    gen = torch.Generator().manual_seed(rank)
    lat = torch.randn(shape, generator=gen).to(dev) * 0.18215
    pe, ue = torch.randn(1, 77, ctx_dim, generator=gen).to(dev), torch.randn(1, 77, ctx_dim, generator=gen).to(dev)
    print(f"[rank {rank}] built in {time.time() - t0:.1f}s, {sync.numel / 1e6:.1f} M LoRA params in one flat buffer", file=sys.stderr, flush=True)

    def step():
        return distill_step(student, teacher, solver, sched, lat, pe, ue, optimizer=opt, grad_sync=sync,
                            autocast_dtype=torch.bfloat16, student_engine=eng, batch_teacher=bool(a.batch_teacher))

    variants = [v for v in a.native_variants.split(",") if v] or [None]
    for variant in variants:
        if variant is not None:  # a fresh engine per variant (its own packs and launch lists) on the same student
            import gc
            eng = None
            gc.collect()
            torch.cuda.empty_cache()
            eng = UNetGradEngine(student, HipOps())
            flags = set(variant.split("+"))
            eng.flash_attn_bwd, eng.tn_wgrad, eng.use_graph = "flash" in flags, "tn" in flags, "graph" in flags
            eng.checkpoint_blocks = "ckpt" in flags      # the reference's use_checkpoint: recompute each block in the backward
            eng.fuse_gn = "nocs" not in flags            # GroupNorm statistics of the forward from the producing GEMMs' epilogues
            eng.group_wgrad = "nogroup" not in flags     # one t2v_wgrad_tn_group launch pair per LoRA group
            eng.bind_lora(params)
            torch.cuda.reset_peak_memory_stats()
        for _ in range(a.warmup + (2 if variant and "graph" in variant else 0)):  # graphs are captured on the second replay
            loss, _ = step()
            print(f"[rank {rank}] warmup loss {float(loss):.4f} ({time.time() - t0:.1f}s)", file=sys.stderr, flush=True)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        marker = torch.zeros(64, device=dev)
        torch.sinh(marker)   # brackets the timed region in a kernel trace (tools/trace_window.py); used nowhere else
        t1 = time.perf_counter()
        for _ in range(a.steps):
            loss, info = step()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / a.steps
        torch.sinh(marker)
        if rank == 0:
            print(json.dumps({"metric": "v1 distillation steps/sec (student " + ("native gradient engine" if eng is not None else "fwd+bwd torch path")
                              + ", teacher x2 native HIP)", "student_native": eng is not None or bool(a.module_route), "variant": variant or ("module-route" if a.module_route else None),
                              "value": round(world / dt, 4), "unit": "samples/s", "n_gpus": world, "ms_per_step": round(dt * 1e3, 1),
                              "lora_grad_mb": round(sync.numel * 4 / 2 ** 20, 1), "loss": float(loss), "host_ms_last_step": info.get("host_ms"),
                              "teacher_native": teacher._engine_box.engine is not None, "peak_mem_gb":
                              round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                              "student_activation_pool_gb": None if eng is None else round(eng.pool.bytes / 2 ** 30, 2)}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
