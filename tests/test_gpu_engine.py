"""-m gpu: the product path end to end on a real MI355X — UNetModel / AutoencoderKL called exactly
like the reference calls them, CUDA tensors in, HIP engine underneath — against the golden vectors
produced by the reference and against the CPU oracle.

Tolerance (stated per BASELINE.md §4): bf16 device path vs fp32 reference, end-to-end rel-L2
<= 3e-2 (the reference's own bf16-vs-fp32 gap is 2.2e-2); attention probabilities (fp32 out of
bf16 q/k) <= 2e-2."""
import os

import pytest
import torch

from oracle import unet_oracle as uo
from oracle.synth import synth_state_dict
from tests.util import VAE_TINY_DD, load, manifest, rel_l2, tiny_unet_params

pytestmark = pytest.mark.gpu
E2E_TOL = 3e-2


def _unet(cfg, man, dtype=torch.float32):
    from t2v_turbo_amd.unet3d import UNetModel
    m = UNetModel(**cfg).eval()
    m.load_state_dict(synth_state_dict(manifest(man)), strict=True)
    return m.to("cuda", dtype)


def test_native_library_is_loaded_and_exports_abi():
    from t2v_turbo_amd import native
    lib = native.load()
    assert lib.t2v_version() >= 100
    assert lib.t2v_init() == 0
    for name in native.EXPORTED:
        assert hasattr(lib, name)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_unet_tiny_vs_reference_golden(dtype):
    """Both parametrisations are the bf16 device path: the engine computes in bf16 (fp32 accumulation) whatever the caller's
    dtype is — fp32 tensors are converted on entry and exit, there is NO fp32 device path (BASELINE.md 4's 5e-4 figure would be
    for one; the fp32 end of the parity chain is the CPU oracle, <= 1e-5 against the reference).  Hence one tolerance, 3e-2."""
    g = load("unet_tiny")
    m = _unet(tiny_unet_params(record_attn_probs=True), "unet_tiny", dtype)
    m.dtype = dtype
    x, ctx, tc = g["x"].cuda().to(dtype), g["ctx"].cuda().to(dtype), g["tc"].cuda().to(dtype)
    with torch.no_grad():
        y = m(x, g["ts"].cuda(), context=ctx, fps=16, timestep_cond=tc)
    assert y.dtype == dtype and y.shape == g["y"].shape
    assert m._engine_box.engine is not None, "CUDA forward must run on the native engine"
    err = rel_l2(y.float().cpu(), g["y"])
    assert err < E2E_TOL, err
    probs = dict(m.named_modules())["output_blocks.11.2.transformer_blocks.0.attn1"].attention_probs
    assert rel_l2(probs.cpu(), g["probs_ob11"]) < 2e-2
    with torch.no_grad():  # teacher-style call
        y2 = m(x, g["ts"].cuda(), context=ctx)
    assert rel_l2(y2.float().cpu(), g["y_nocond"]) < E2E_TOL


def test_unet_tiny_motion_cond_batch2_replay_and_graph():
    g = load("unet_tiny_mg_b2")
    m = _unet(tiny_unet_params(motion_cond_proj_dim=256), "unet_tiny_mg_b2")
    args = (g["x"].cuda(), g["ts"].cuda())
    kw = dict(context=g["ctx"].cuda(), fps=8, timestep_cond=g["tc"].cuda(), motion_cond=g["mc"].cuda())
    with torch.no_grad():
        y_rec = m(*args, **kw)          # recording pass
        y_rep = m(*args, **kw)          # replay of the recorded launches
    assert rel_l2(y_rec.cpu(), g["y"]) < E2E_TOL
    assert torch.equal(y_rec, y_rep), "replay must be bit-identical to the recording pass"
    eng = m.native_engine()
    eng.use_graph = True
    with torch.no_grad():
        y_g1 = m(*args, **kw)           # captures the hipGraph
        y_g2 = m(*args, **kw)           # graph replay
    plan = next(iter(eng.plans.values()))
    assert plan.get("graph") is not None, plan.get("graph_failed")
    assert torch.equal(y_g1, y_rec) and torch.equal(y_g2, y_rec)
    # new inputs through the graph
    x2 = torch.randn(g["x"].shape, generator=torch.Generator().manual_seed(3))
    ts2 = torch.tensor([759, 19])
    with torch.no_grad():
        y3 = m(x2.cuda(), ts2.cuda(), **kw)
    sd = synth_state_dict(manifest("unet_tiny_mg_b2"))
    ref = uo.unet_forward(sd, tiny_unet_params(motion_cond_proj_dim=256), x2, ts2, g["ctx"], fps=8,
                          timestep_cond=g["tc"], motion_cond=g["mc"])
    assert rel_l2(y3.cpu(), ref) < E2E_TOL


def test_unet_c1_shape_family_vs_oracle():
    """config C1 latent shape (1,4,8,32,32) on the tiny-width model: exercises 1024/256/64/16-token levels."""
    cfg = tiny_unet_params()
    sd = synth_state_dict(manifest("unet_tiny"))
    m = _unet(cfg, "unet_tiny")
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(1, 4, 8, 32, 32, generator=gen)
    ctx = torch.randn(1, 77, cfg["context_dim"], generator=gen)
    tc = torch.randn(1, 256, generator=gen)
    ts = torch.tensor([519])
    ref = uo.unet_forward(sd, cfg, x, ts, ctx, fps=16, timestep_cond=tc)
    with torch.no_grad():
        y = m(x.cuda(), ts.cuda(), context=ctx.cuda(), fps=16, timestep_cond=tc.cuda())
    assert rel_l2(y.cpu(), ref) < E2E_TOL


def test_vae_decode_vs_reference_golden():
    from t2v_turbo_amd.vae import AutoencoderKL
    g = load("vae_tiny")
    ae = AutoencoderKL(ddconfig=VAE_TINY_DD, embed_dim=4).eval()
    ae.load_state_dict(synth_state_dict(manifest("vae_tiny")), strict=True)
    ae = ae.cuda()
    with torch.no_grad():
        v = ae.decode_video(g["z"].cuda())
        v2 = ae.decode_video(g["z"].cuda())
    assert v.shape == g["video"].shape
    assert ae._engine_box.engine is not None
    assert rel_l2(v.cpu(), g["video"]) < E2E_TOL
    assert torch.equal(v, v2)
    with torch.no_grad():  # single-frame AutoencoderKL.decode API
        f0 = ae.decode(g["z"][:, :, 0].cuda() / 0.18215)
    assert rel_l2(f0.cpu(), g["video"][:, :, 0]) < E2E_TOL


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from t2v_turbo_amd import native
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setenv("T2V_HIP_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(native.NativeError):
        native.load()


def test_pipeline_on_gpu_vs_reference_pipeline_fixture():
    """The reference pipeline's own output (tiny widths, CPU fp32) vs our pipeline on the GPU:
    HIP UNet x4 + fused scheduler step + batched HIP VAE decode.  Noise is drawn from the same CPU
    generator, so the trajectories are comparable; tolerance = 4 accumulated bf16 UNet steps."""
    from t2v_turbo_amd.latent_diffusion import LatentDiffusion
    from t2v_turbo_amd.pipeline import T2VTurboVC2Pipeline
    from t2v_turbo_amd.scheduler import T2VTurboScheduler
    from t2v_turbo_amd.unet3d import UNetModel
    from t2v_turbo_amd.vae import AutoencoderKL
    g = load("pipeline_tiny")
    p = tiny_unet_params()
    unet = UNetModel(**p).eval()
    unet.load_state_dict(synth_state_dict(manifest("unet_tiny")), strict=True)
    ae = AutoencoderKL(ddconfig=VAE_TINY_DD, embed_dim=4).eval()
    ae.load_state_dict(synth_state_dict(manifest("vae_tiny")), strict=True)
    t2v = LatentDiffusion(unet, ae).cuda()
    pipe = T2VTurboVC2Pipeline(t2v, T2VTurboScheduler(), {"params": {"unet_config": {"params": p}}})
    kw = dict(prompt=None, height=64, width=64, frames=4, fps=16, guidance_scale=7.5, num_inference_steps=4,
              lcm_origin_steps=50, prompt_embeds=g["prompt_embeds"].cuda())
    lat = pipe(generator=torch.Generator().manual_seed(42), output_type="latent", **kw)
    vid = pipe(generator=torch.Generator().manual_seed(42), output_type="pt", **kw)
    assert unet._engine_box.engine is not None and ae._engine_box.engine is not None
    assert rel_l2(lat.cpu(), g["latent"]) < 6e-2
    assert vid.shape == g["video"].shape and rel_l2(vid.cpu(), g["video"]) < 8e-2


def test_scheduler_step_fused_kernel_matches_torch():
    from t2v_turbo_amd.scheduler import T2VTurboScheduler
    g = load("sched")
    s = T2VTurboScheduler()
    s.set_timesteps(4, 50)
    for i, t in enumerate(s.timesteps):
        gen = torch.Generator().manual_seed(100 + i)
        with torch.no_grad():
            prev, den = s.step(g["mout"].cuda(), i, t, g["sample"].cuda(), generator=gen, return_dict=False)
        assert rel_l2(prev.cpu(), g[f"prev_{i}"]) < 1e-5 and rel_l2(den.cpu(), g[f"den_{i}"]) < 1e-5


def test_vae_encode_vs_reference_golden():
    from t2v_turbo_amd.vae import AutoencoderKL
    g = load("vae_tiny_enc")
    ae = AutoencoderKL(ddconfig=VAE_TINY_DD, embed_dim=4).eval()
    ae.load_state_dict(synth_state_dict(manifest("vae_tiny")), strict=True)
    ae = ae.cuda()
    with torch.no_grad():
        post = ae.encode(g["x"].cuda())
        mom = ae.encode_moments_video(g["x"].cuda().unsqueeze(0).transpose(1, 2).contiguous())
    assert ae._engine_box.enc is not None
    assert rel_l2(post.parameters.cpu(), g["moments"]) < E2E_TOL
    assert rel_l2(post.mean.cpu(), g["mean"]) < E2E_TOL
    assert rel_l2(mom[0].transpose(0, 1).cpu(), g["moments"]) < E2E_TOL


def test_flat_adamw_and_ema_kernels_match_torch():
    from tests.test_optim import _run
    from t2v_turbo_amd.optim import update_ema_flat
    _run("cuda")
    t, s = torch.randn(100003, device="cuda"), torch.randn(100003, device="cuda")
    ref = t * 0.95 + s * 0.05
    update_ema_flat(t, s, 0.95)
    assert torch.allclose(t, ref, rtol=1e-6, atol=1e-7)


def test_vae_decode_backward_native_vs_oracle_autograd():
    """d(loss)/d(latents) through vae.decode on the HIP engine (forward + backward launch lists) vs
    torch autograd through the pinned fp32 oracle decoder; also through the public ``AutoencoderKL.decode`` autograd path."""
    from oracle import synth, vae_oracle
    from t2v_turbo_amd.vae import AutoencoderKL
    from tests.util import VAE_TINY_DD
    dd = dict(VAE_TINY_DD)
    ae = AutoencoderKL(ddconfig=dd, embed_dim=4).eval()
    sd = synth.synth_state_dict(synth.manifest_of(ae))
    ae.load_state_dict(sd)
    ae = ae.cuda().bfloat16().requires_grad_(False)
    g = torch.Generator().manual_seed(3)
    z = torch.randn(3, 4, 16, 16, generator=g)
    dout = torch.randn(3, 3, 128, 128, generator=g)
    sd32 = {k: v.float() for k, v in sd.items()}
    zz = z.bfloat16().float().clone().requires_grad_(True)
    with torch.enable_grad():
        ref = vae_oracle.decoder_forward.__wrapped__(sd32, dd, torch.nn.functional.conv2d(
            zz, sd32["post_quant_conv.weight"], sd32["post_quant_conv.bias"]))
        (ref * dout.bfloat16().float()).sum().backward()
    zc = z.cuda().bfloat16().requires_grad_(True)
    out = ae.decode(zc)
    assert ae._engine_box.grad is not None and ae._engine_box.grad.ops.is_native
    (out.float() * dout.cuda().bfloat16().float()).sum().backward()
    assert rel_l2(out.float().cpu(), ref.detach()) < 3e-2
    err = rel_l2(zc.grad.float().cpu(), zz.grad)
    assert err < 5e-2, err  # bf16 activations and gradients end to end (forward tolerance is 3e-2)
    # second call replays both launch lists
    zc2 = (z * 0.5).cuda().bfloat16().requires_grad_(True)
    out2 = ae.decode(zc2)
    (out2.float() * dout.cuda().bfloat16().float()).sum().backward()
    zz2 = (z * 0.5).bfloat16().float().clone().requires_grad_(True)
    with torch.enable_grad():
        ref2 = vae_oracle.decoder_forward.__wrapped__(sd32, dd, torch.nn.functional.conv2d(
            zz2, sd32["post_quant_conv.weight"], sd32["post_quant_conv.bias"]))
        (ref2 * dout.bfloat16().float()).sum().backward()
    assert rel_l2(zc2.grad.float().cpu(), zz2.grad) < 5e-2


def test_unet_full_width_c2_config_vs_oracle():
    """Parity where the metric is quoted (BASELINE configs[1]): VideoCrafter2 widths, latent (1,4,16,40,64), bf16 device path,
    hipGraph replay included, against the fp32 CPU oracle on the same weights (zero-init tensors re-drawn) — <= 3e-2."""
    import bench
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev, torch.bfloat16)
    x, ctx, tc = bench.synth_inputs(dev, torch.bfloat16)
    ts = torch.tensor([999], device=dev)
    eng = model.native_engine()
    eng.use_graph = True
    with torch.no_grad():
        ys = [model(x, ts, context=ctx, fps=16, timestep_cond=tc).float().cpu() for _ in range(3)]  # record, replay, graph
    assert model._engine_box.engine is not None and torch.isfinite(ys[0]).all()
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2]), "replays must be bit-identical to the recording pass"
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    ref = uo.unet_forward(sd, bench.VC2_UNET, x.float().cpu(), ts.cpu(), ctx.float().cpu(), fps=16, timestep_cond=tc.float().cpu())
    err = rel_l2(ys[0], ref)
    print(f"full-width C2 parity: rel-L2 {err:.3e}")
    assert err < E2E_TOL, err
    # the default dataflow takes GroupNorm statistics from the producing GEMMs and folds the 16 text cross-attention LayerNorms into their q GEMMs;
    # switched off (T2V_FUSE_GN=0 / T2V_FOLD_LN=0 in the environment, or the engine attributes) it is the round-2 network: same
    # numbers within two bf16 roundings; and the opt-in LayerNorm-as-second-output form on top of that (30 launches fewer)
    n_default = len(next(iter(eng.plans.values()))["rec"])
    outs = {}
    # (ln_in_fill off in both: the LayerNorms t2v_linear_pr takes into its panel fill would otherwise be gone from both counts)
    for tag, flags in (("plain", dict(fuse_gn=False, fold_ln=False, fuse_ln=False, ln_in_fill=False)),
                       ("ln_second_output", dict(fuse_gn=False, fold_ln=False, fuse_ln=True, ln_in_fill=False))):
        for k, v in flags.items():
            setattr(eng, k, v)
        eng.plans.clear()
        try:
            with torch.no_grad():
                outs[tag] = (model(x, ts, context=ctx, fps=16, timestep_cond=tc).float().cpu(), len(next(iter(eng.plans.values()))["rec"]))
        finally:
            eng.fuse_gn = eng.fold_ln = eng.ln_in_fill = True
            eng.fuse_ln = False
            eng.plans.clear()
    n_plain = outs["plain"][1]
    print(f"launches per step: default {n_default}, plain {n_plain}, LayerNorm as second output {outs['ln_second_output'][1]}")
    assert n_default < n_plain and outs["ln_second_output"][1] == n_plain - 30
    for tag, (y_v, _) in outs.items():
        assert rel_l2(y_v, ref) < E2E_TOL and rel_l2(y_v, ys[0]) < E2E_TOL, tag   # bf16 roundings of one network: each ~1.7e-2 from fp32


def test_unet_full_width_c1_geometry_off_the_tuned_table():
    """A geometry the tile table was NOT tuned on (VERDICT r4 "what's weak" 9): BASELINE configs[0]'s latent (1,4,8,32,32) — 8 frames
    of 256x256 — at the full VC2 widths.  gemm_tune.json holds exactly the bench shapes (M = 16 * 40 * 64 / 4^level); here almost no
    t2v_gemm launch has an exact entry (only the token-count-independent ones: text K / V, time embedding) and the halo kernel tiles
    a 32x32 image with 10-row tiles (80 % of its rows).  Parity against the fp32 oracle at the same tolerance, and the step is timed
    three ways so that the distance between tuned and untuned is a number, not a guess: exact table (bench geometry), nearest-shape
    fallback (``native.TuneTable``: same (mode, N, K), nearest M), and the bare library heuristic."""
    import bench
    from t2v_turbo_amd.native import TuneTable
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev, torch.bfloat16)
    eng = model.native_engine()
    eng.use_graph = True
    table = eng.ops.tune
    assert isinstance(table, TuneTable) and len(table) > 300
    x16, ctx, tc = bench.synth_inputs(dev, torch.bfloat16)
    x8 = torch.randn(1, 4, 8, 32, 32, generator=torch.Generator().manual_seed(3)).to(dev, torch.bfloat16)
    ts = torch.tensor([759], device=dev)

    def run(x, reps=8):
        eng.plans.clear()
        for k in table.stats:
            table.stats[k] = 0
        with torch.no_grad():
            ys = [model(x, ts, context=ctx, fps=16, timestep_cond=tc) for _ in range(3)]   # record, replay, graph capture
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                model(x, ts, context=ctx, fps=16, timestep_cond=tc)
            e1.record()
            torch.cuda.synchronize()
        assert torch.equal(ys[0], ys[2])
        return ys[0].float().cpu(), e0.elapsed_time(e1) / reps, dict(table.stats)

    y8n, ms8n, st8n = run(x8)                 # nearest-shape fallback (the default)
    table.nearest = False
    try:
        y8h, ms8h, st8h = run(x8)             # exact keys only: the library heuristic everywhere else
    finally:
        table.nearest = True
    y16, ms16, st16 = run(x16)
    print(f"tile-table lookups: C1 geometry {st8n} (nearest-shape on) / {st8h} (exact only); bench geometry {st16}", flush=True)
    assert st8n["exact"] < 60 and st8n["nearest"] > 200, st8n      # this geometry is not in the table; its (mode, N, K) families are
    assert st8h["nearest"] == 0 and st8h["miss"] > 200, st8h
    assert st16["exact"] > 300 and st16["exact"] > 8 * (st16["nearest"] + st16["miss"]), st16
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    ref = uo.unet_forward(sd, bench.VC2_UNET, x8.float().cpu(), ts.cpu(), ctx.float().cpu(), fps=16, timestep_cond=tc.float().cpu())
    err_n, err_h = rel_l2(y8n, ref), rel_l2(y8h, ref)
    # 2.442 TFLOP at 8x32x32, 12.581 TFLOP at 16x40x64 (2 MAC, GEMM + conv + attention)
    tf8n, tf8h, tf16 = 2.442 / ms8n * 1e3, 2.442 / ms8h * 1e3, 12.581 / ms16 * 1e3
    print(f"C1 geometry (1,4,8,32,32): nearest-shape tiles {ms8n:.2f} ms = {tf8n:.0f} TFLOP/s (rel-L2 {err_n:.3e}, lookups {st8n}); "
          f"library heuristic {ms8h:.2f} ms = {tf8h:.0f} TFLOP/s (rel-L2 {err_h:.3e}, lookups {st8h}); "
          f"bench geometry on its tuned table {ms16:.2f} ms = {tf16:.0f} TFLOP/s (lookups {st16})", flush=True)
    assert err_n < E2E_TOL and err_h < E2E_TOL, (err_n, err_h)
    # a fifth of the tokens per launch: the small geometry runs at a lower rate even on good tiles (fewer tiles than CUs at the lower
    # levels); the bounds only catch a tile choice that falls off a cliff, and a fallback that is worse than no fallback
    assert tf8n > 0.25 * tf16 and ms8n < 1.10 * ms8h, (tf8n, tf16, ms8n, ms8h)   # (measured: 4-5 % FASTER; 10 % of slack for a noisy box)


def test_unet_full_width_motion_cond_config_c4_vs_oracle():
    """BASELINE configs[3] (T2V-Turbo-v2 sampling): the motion-conditioned UNet (`unet_mg`: motion_cond_proj_dim = 256,
    pipeline/t2v_turbo_vc2_pipeline.py:190-204) at the VC2 widths on latent (1,4,16,40,64), bf16 device path vs the fp32 CPU
    oracle, at a timestep of the 16-step grid above the motion threshold (t >= 700): the forward bench.py's clip_16step_v2 runs."""
    import bench
    from t2v_turbo_amd.nn_util import guidance_embedding
    from t2v_turbo_amd.unet3d import UNetModel
    dev = torch.device("cuda", 0)
    cfg = dict(bench.VC2_UNET, motion_cond_proj_dim=256)
    torch.manual_seed(1234)
    with torch.device(dev):
        model = UNetModel(**cfg)
    g = torch.Generator(device=dev).manual_seed(1234)
    with torch.no_grad():
        for p in model.parameters():
            if float(p.abs().max()) == 0.0:
                p.normal_(0.0, 0.02, generator=g)
    model = model.to(torch.bfloat16).eval()
    model.dtype = torch.bfloat16
    x, ctx, tc = bench.synth_inputs(dev, torch.bfloat16)
    mc = guidance_embedding(torch.tensor([0.1]), 256).to(dev, torch.bfloat16)
    ts = torch.tensor([939], device=dev)
    with torch.no_grad():
        ys = [model(x, ts, context=ctx, fps=16, timestep_cond=tc, motion_cond=mc).float().cpu() for _ in range(2)]
        y_off = model(x, ts, context=ctx, fps=16, timestep_cond=tc,
                      motion_cond=guidance_embedding(torch.tensor([0.0]), 256).to(dev, torch.bfloat16)).float().cpu()
    assert model._engine_box.engine is not None and torch.isfinite(ys[0]).all() and torch.equal(ys[0], ys[1])
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    ref = uo.unet_forward(sd, cfg, x.float().cpu(), ts.cpu(), ctx.float().cpu(), fps=16, timestep_cond=tc.float().cpu(),
                          motion_cond=mc.float().cpu())
    err = rel_l2(ys[0], ref)
    print(f"full-width C4 (motion cond) parity: rel-L2 {err:.3e}; motion embedding on vs off: {rel_l2(y_off, ys[0]):.3e}")
    assert err < E2E_TOL, err
    assert rel_l2(y_off, ys[0]) > 1e-3, "the motion-guidance embedding must reach the output"


def test_rccl_world_size_1_flat_gradient_all_reduce():
    """The RCCL path of dist.py (backend "nccl" = RCCL on ROCm) on the one GPU there is: a one-rank communicator, one
    all-reduce of the v1 LoRA gradient buffer (117.1 M fp32 = 468.6 MB, train_t2v_turbo_v1_lora.py:1190), mean semantics.
    (In a child interpreter: tests.util.run_isolated.)"""
    from tests.util import run_isolated
    run_isolated("tests.test_gpu_engine", "_body_rccl_world_size_1_flat_gradient_all_reduce")


def _body_rccl_world_size_1_flat_gradient_all_reduce():
    import torch.distributed as dist
    from t2v_turbo_amd import dist as tdist
    assert not dist.is_initialized()
    tdist.init_distributed("nccl", single_process_group=True)
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        p = torch.nn.Parameter(torch.zeros(117_150_000, device="cuda"))
        sync = tdist.FlatGradSync([p])
        sync.flat.copy_(torch.arange(sync.numel, device="cuda", dtype=torch.float32) % 1000)
        want = sync.flat.clone()
        work = sync.all_reduce_mean(force=True)
        assert work is None or work is not None  # sync call: completed on return
        torch.cuda.synchronize()
        assert torch.equal(sync.flat, want)
        gathered = [torch.zeros(3, device="cuda")]
        dist.all_gather(gathered, torch.tensor([1.0, 2.0, 3.0], device="cuda"))  # the logged-loss exchange
        assert gathered[0].tolist() == [1.0, 2.0, 3.0]
        norm = sync.clip_grad_norm_(1.0)
        assert torch.isfinite(norm) and abs(float(sync.flat.norm()) - 1.0) < 1e-3
        torch.cuda.synchronize()
        print("BODY_OK", flush=True)
    finally:
        dist.destroy_process_group()


def test_ddim_inversion_and_motion_prior_score_on_device():
    """SURVEY 8(f) rank 4 on the GPU: DDIM inversion (motion_prior_sample.py:27-37) as a consumer of the native UNet forward, and
    ``get_motion_prior_score`` (:59-84) on the data-gradient engine against the score the REFERENCE computed
    (tests/golden/unet_tiny_grad.npz)."""
    from t2v_turbo_amd import cd_math, motion_prior as mp
    from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
    from t2v_turbo_amd.native import HipOps
    from t2v_turbo_amd.scheduler import T2VTurboScheduler
    g, gg = load("unet_tiny"), load("unet_tiny_grad")
    m = _unet(tiny_unet_params(record_attn_probs=True), "unet_tiny", torch.bfloat16)
    m.dtype = torch.bfloat16
    m.requires_grad_(False)
    dev = torch.device("cuda", 0)
    # inversion: 3 reverse steps on the native engine, each undone by the forward DDIM step with the same prediction
    sched = T2VTurboScheduler(linear_start=0.00085, linear_end=0.012)
    solver = cd_math.DDIMSolver(sched.alphas_cumprod.numpy(), ddim_timesteps=50).to(dev)
    ctx = {"context": g["ctx"].cuda().bfloat16(), "fps": 16, "timestep_cond": g["tc"].cuda().bfloat16()}
    x0 = g["x"].cuda().bfloat16()
    lat = mp.reverse_ddim_loop(x0, m, ctx, solver, 3, dev)
    assert m._engine_box.engine is not None and len(lat) == 3 and all(torch.isfinite(t).all() for t in lat)
    ts0 = solver.ddim_timesteps[torch.tensor([0], device=dev)].long()
    with torch.no_grad():
        eps = m(x0, ts0, **ctx).double().cpu()
    assert torch.isfinite(eps).all()
    acp = solver.alpha_cumprods.double().cpu()
    a_t, a_prev = acp[int(ts0)], acp[max(int(ts0) - solver.step_ratio, 0)]
    x_back = a_prev.sqrt() * (lat[0].double().cpu() - (1 - a_t).sqrt() * eps) / a_t.sqrt() + (1 - a_prev).sqrt() * eps
    assert rel_l2(x_back.float(), x0.float().cpu()) < 2e-2   # bf16 latents between the steps
    # motion-prior score on the gradient engine vs the reference's own number
    eng = UNetGradEngine(m, HipOps())
    cctx = {"context": g["ctx"].cuda().bfloat16(), "fps": 16, "timestep_cond": g["tc"].cuda().bfloat16()}
    score, out = mp.get_motion_prior_score_native(eng, m, g["x"].cuda().bfloat16(), g["ts"].cuda(), gg["example"].cuda().bfloat16(),
                                                  cctx, cctx, 500.0)
    assert rel_l2(out.float().cpu(), gg["out"]) < E2E_TOL
    assert torch.isfinite(score).all()
    assert rel_l2(score.float().cpu(), gg["score"]) < 0.15   # bf16 probabilities through a top-1 selection loss (fp32 path: 1e-4, CPU suite)


def test_alternating_input_signatures_keep_their_plans():
    """Two recorded plans of one engine (same weights, different input signature) must both stay replayable: a plan owns the
    buffers its launch list points at (regression: recording the second plan freed the first one's pool)."""
    g = load("unet_tiny")
    m = _unet(tiny_unet_params(), "unet_tiny", torch.bfloat16)
    m.dtype = torch.bfloat16
    x, ctx, tc = g["x"].cuda().bfloat16(), g["ctx"].cuda().bfloat16(), g["tc"].cuda().bfloat16()
    ts = g["ts"].cuda()
    with torch.no_grad():
        a0 = m(x, ts, context=ctx, fps=16, timestep_cond=tc)            # plan A (recording)
        b0 = m(x.float(), ts, context=ctx.float(), fps=16)              # plan B: fp32 inputs, no guidance embedding
        big = [torch.randn(1 << 22, device="cuda") for _ in range(8)]  # allocator traffic in between
        a1 = m(x, ts, context=ctx, fps=16, timestep_cond=tc)            # back to plan A (replay)
        b1 = m(x.float(), ts, context=ctx.float(), fps=16)
        a2 = m(x, ts, context=ctx, fps=16, timestep_cond=tc)
    del big
    assert len(m._engine_box.engine.plans) == 2
    assert torch.isfinite(a1).all() and torch.equal(a0, a1) and torch.equal(a0, a2) and torch.equal(b0.float(), b1.float())
    assert rel_l2(a1.float().cpu(), g["y"]) < E2E_TOL


def test_train_mode_frozen_teacher_on_device_no_warning_and_mask_replay():
    """The v1 teacher's call pattern, unchanged: a frozen LoRA-free network that was never put in eval mode, called under no_grad
    (train_t2v_turbo_v1_lora.py:621-626,1105-1134).  It must land on the native engine without the composite-path RuntimeWarning;
    the TemporalConvBlock dropouts are the counter-based device masks (regenerated on the host from seed / site / geometry: bit-identical,
    tests/emu_ops.py::dropout_keep) and replaying them inside the torch module reproduces the device output; hipGraph replay follows
    the per-call seed."""
    import copy
    import warnings
    from tests.emu_ops import EmuOps
    from tests.mask_replay import patch_engine_masks
    g = load("unet_tiny")
    m = _unet(tiny_unet_params(), "unet_tiny")
    m.requires_grad_(False)
    m.train()
    eng = m.native_engine()
    eng.seed_source = iter([77, 78, 78, 79, 79])
    x, ts, ctx = g["x"].cuda(), g["ts"].cuda(), g["ctx"].cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("error")           # the ATen composite route announces itself with a RuntimeWarning
        with torch.no_grad():
            y = m(x, ts, context=ctx)
    assert eng.drop_sites and all(kind == "tconv" for _, kind, _ in eng.drop_sites)
    assert rel_l2(y.float().cpu(), g["y_nocond"]) > 1e-3    # masks applied
    # host replay of the masks of seed 77 inside the fp32 torch module
    masks = {}
    ref_m = copy.deepcopy(m).float().cpu()
    widths = {}
    for blk in ref_m.modules():
        if type(blk).__name__ == "TemporalConvBlock":
            for stg in (blk.conv1, blk.conv2, blk.conv3, blk.conv4):
                for layer in stg:
                    if isinstance(layer, torch.nn.Dropout):
                        widths[id(layer)] = stg[0].num_channels
    twin = {id(a): b for a, b in zip(m.modules(), ref_m.modules())}
    for sid, (drops, _, (B, F, h, w)) in enumerate(eng.drop_sites):
        C = widths[id(twin[id(drops[0])])]
        masks[sid] = EmuOps.dropout_keep(77, sid, B * F * h * w, C, drops[0].p)
    ref_eng_like = type("E", (), {"model": m, "drop_sites": eng.drop_sites})()
    patch_engine_masks(ref_m, ref_eng_like, masks)
    ref_m.native_mode = "off"
    with torch.no_grad():
        ref = ref_m(g["x"], g["ts"], context=g["ctx"])
    assert rel_l2(y.float().cpu(), ref) < E2E_TOL
    # replayed launch list: a new seed per call, the same seed -> bit-identical; then as one hipGraph
    with torch.no_grad():
        y2 = m(x, ts, context=ctx)
        y3 = m(x, ts, context=ctx)
    assert torch.equal(y2, y3) and not torch.equal(y2, y)
    eng.use_graph = True
    with torch.no_grad():
        y4 = m(x, ts, context=ctx)      # seed 79: captures
        y5 = m(x, ts, context=ctx)      # seed 79: graph replay
    assert torch.equal(y4, y5) and not torch.equal(y4, y2)


def test_vae_decode_full_size_vs_oracle():
    """The KL-VAE decoder at the size the clip legs time it on (ae_modules.py:29-73,602-641; ddpm3d.py:666-679): ch = 128, ch_mult
    (1,2,4,4), latent 40x64 -> 320x512 incl. the single-head 2 560-token AttnBlock at width 512; two frames, bf16 engine against the
    fp32 oracle on the same random-init weights.  Tolerance: BASELINE.md 4's end-to-end 3e-2."""
    from oracle import vae_oracle as vo
    from t2v_turbo_amd.vae import AutoencoderKL
    from tests.util import VAE_FULL_DD
    torch.manual_seed(7)
    ae = AutoencoderKL(ddconfig=VAE_FULL_DD, embed_dim=4).eval()
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in ae.parameters():   # (zero-init tensors re-drawn so that every branch carries signal)
            if float(p.abs().max()) == 0.0:
                p.normal_(0.0, 0.02, generator=gen)
    sd = {k: v.detach().clone() for k, v in ae.state_dict().items()}
    z = torch.randn(1, 4, 2, 40, 64, generator=gen) * 0.18215 * 4.0
    ref = vo.decode_first_stage_2dae(sd, VAE_FULL_DD, z)
    ae_dev = ae.to("cuda", torch.bfloat16)
    with torch.no_grad():
        v = ae_dev.decode_video(z.cuda().bfloat16())
    assert ae_dev._engine_box.engine is not None, "native VAE engine did not run"
    assert v.shape == ref.shape == (1, 3, 2, 320, 512)
    err = rel_l2(v.float().cpu(), ref)
    print(f"[full-size VAE decode] rel-L2 vs fp32 oracle {err:.3e}", flush=True)
    assert err < E2E_TOL, err
