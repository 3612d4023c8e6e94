"""-m gpu: hardware parity of the LoRA TRAINING path at the size it is timed on, and of the route the trainer takes.

The distillation step `bench.py` times (`distill_step` leg: VideoCrafter2 widths, latent (1,4,16,40,64), LoRA r = 64) runs the
student on the native gradient engine.  These tests gate that path on MI355X the way `test_gpu_engine.py` gates inference:

  (i)   full-width student forward + backward on the device vs fp32 CPU autograd through the torch module
        (utils/lora.py:45-50,124-129,204-209 forward; train_t2v_turbo_v1_lora.py:1022-1028,1190 backward): output, d/d(latents)
        and EVERY LoRA tensor's gradient by cosine and norm ratio (a scale slip on one leaf fails);
  (ii)  tiny width DIRECTLY against tests/golden/unet_tiny_lora_grad.npz — gradients the reference itself computed;
  (iii) the trainer's route: `unet(...)` -> `_NativeStudent` -> `loss.backward()` inside `distill.distill_step`, two steps with a
        `FlatAdamW` update between them; the second step must have seen the update (operand packs refreshed);
  (iv)  train mode: the engine's counter-based dropout masks replayed inside the torch module (tests/mask_replay.py).

Tolerances (bf16 device path vs fp32 reference): output <= 3e-2 rel-L2, d/d(latents) <= 6e-2, LoRA gradients per tensor
cosine >= 0.99 and norm within 10 % (tiny width: >= 0.985 — 64-channel GroupNorm groups of 2 channels are noisier)."""
import os
import time

import pytest
import torch

from tests.util import load, manifest, rel_l2, tiny_unet_params

pytestmark = pytest.mark.gpu

OUT_TOL, DX_TOL = 3e-2, 6e-2


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))


def _per_tensor(params, flat, cond_grads, refs, names):
    """[(name, cosine, norm ratio, ref norm)] per LoRA tensor: engine leaves from the flat buffer, conditioning leaves from torch."""
    rows, off = [], 0
    for p, r in zip(params, refs):
        g = flat[off:off + p.numel()].view_as(p).float().cpu()
        off += p.numel()
        cg = cond_grads.get(id(p))
        if cg is not None:
            g = g + cg.float().cpu()
        rn = float(r.double().norm())
        if rn == 0.0:
            assert float(g.abs().max()) < 1e-6, names[id(p)]
            continue
        rows.append((names[id(p)], _cos(g, r), float(g.double().norm()) / rn, rn))
    return rows


def _report(tag, rows, cos_min, ratio_tol):
    import statistics
    cs = [r[1] for r in rows]
    rt = [abs(r[2] - 1.0) for r in rows]
    worst = sorted(rows, key=lambda r: r[1])[:5]
    print(f"[{tag}] {len(rows)} gradient tensors: cosine min {min(cs):.4f} median {statistics.median(cs):.5f}; "
          f"|norm ratio - 1| max {max(rt):.3f} median {statistics.median(rt):.4f}; worst: "
          + ", ".join(f"{n} cos {c:.4f} ratio {q:.3f}" for n, c, q, _ in worst), flush=True)
    bad = [r for r in rows if r[1] < cos_min or abs(r[2] - 1.0) > ratio_tol]
    assert not bad, bad[:8]


# ------------------------------------------------------------------------------------------------------------- (i)
def test_student_full_width_forward_backward_vs_cpu_autograd():
    """VC2 widths, latent (1,4,16,40,64), r = 64, eval mode: the configuration of bench.py's distill_step leg."""
    import bench
    from t2v_turbo_amd import lora
    from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
    from t2v_turbo_amd.native import HipOps
    from t2v_turbo_amd.unet3d import UNetModel
    t0 = time.time()
    dev = torch.device("cuda", 0)
    torch.manual_seed(4321)                 # the default initialisers draw from the global generators: without this the weights (and
    torch.cuda.manual_seed_all(4321)        # with them the measured errors, 1.9-2.4e-2 / 5.6-5.9e-2) depend on which tests ran before
    with torch.device(dev):
        student = UNetModel(**bench.VC2_UNET)
    g = torch.Generator(device=dev).manual_seed(4321)
    with torch.no_grad():
        for p in student.parameters():
            if float(p.abs().max()) == 0.0:
                p.normal_(0.0, 0.02, generator=g)
    student.requires_grad_(False)
    lora.inject_trainable_lora_extended(student, r=64)
    params = lora.lora_parameters(student)
    with torch.no_grad():
        for p in params:   # lora_up starts at zero: every lora_down gradient would be zero
            if float(p.abs().max()) == 0.0:
                p.normal_(0.0, 0.02, generator=g)
    student.eval()
    assert len(params) == 1150 and sum(p.numel() for p in params) == 117_142_176  # the v1 all-reduce payload
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 16, 40, 64, generator=gen)
    ctx = torch.randn(1, 77, 1024, generator=gen)
    tc = torch.randn(1, 256, generator=gen)
    r_out = torch.randn(x.shape, generator=gen)
    ts = torch.tensor([519])
    # ---- device: forward + backward on the gradient engine ------------------------------------------------------------
    eng = UNetGradEngine(student, HipOps())
    eng.bind_lora(params)
    ys, dxs, flats = [], [], []
    for rep in range(2):   # recording pass, then the replayed launch lists
        emb_all = student.conditioning_emb_all(ts.to(dev), 16, tc.to(dev))
        y = eng.forward_tape(x.to(dev), ts.to(dev), ctx.to(dev), 16, tc.to(dev), None, emb_all=emb_all)
        flat = torch.zeros(eng.lora_numel, device=dev)
        dx = eng.backward(r_out.to(dev), flat_grad=flat, accumulate=False)
        for p in params:
            p.grad = None
        emb_all.backward(eng.d_emb_all)
        ys.append(y.float().cpu()); dxs.append(dx.float().cpu()); flats.append(flat.cpu())
    cond = {id(p): p.grad.detach().clone() for p in params if p.grad is not None}
    assert torch.isfinite(ys[0]).all() and torch.isfinite(dxs[0]).all() and torch.isfinite(flats[0]).all()
    assert rel_l2(ys[1], ys[0]) < 1e-6 and rel_l2(dxs[1], dxs[0]) < 1e-6 and rel_l2(flats[1], flats[0]) < 1e-5
    print(f"device side done (+{time.time() - t0:.0f}s): {len(eng._last['rec'])} forward / {len(eng._last['rec_bwd'])} backward launches",
          flush=True)
    # ---- host: fp32 autograd through the torch module (oracle/lora_grad_oracle.py; checkpointing bounds the host memory) ----
    from oracle.lora_grad_oracle import student_reference
    y_ref, dx_ref, g_ref = student_reference(student.state_dict(), bench.VC2_UNET, 64, x, ts, ctx, 16, tc, r_out,
                                             threads=min(os.cpu_count() or 1, 64))
    print(f"host reference done (+{time.time() - t0:.0f}s)", flush=True)
    e_out, e_dx = rel_l2(ys[0], y_ref), rel_l2(dxs[0], dx_ref)
    print(f"[full width] out rel-L2 {e_out:.3e}  d/d(latents) rel-L2 {e_dx:.3e}", flush=True)
    names = {id(p): n for n, p in student.named_parameters()}
    rows = _per_tensor(params, flats[0], cond, g_ref, names)
    _report("full width", rows, 0.99, 0.10)
    assert e_out < OUT_TOL and e_dx < DX_TOL, (e_out, e_dx)


# ------------------------------------------------------------------------------------------------------------- (ii)
def _tiny_student(rank, draw, **cfg):
    from oracle.synth import synth_state_dict
    from t2v_turbo_amd import lora
    from t2v_turbo_amd.unet3d import UNetModel
    m = UNetModel(**tiny_unet_params(**cfg)).eval()
    m.load_state_dict(synth_state_dict(manifest("unet_tiny")), strict=True)
    m.requires_grad_(False)
    lora.inject_trainable_lora_extended(m, r=rank)
    params = lora.lora_parameters(m)
    draw(params)
    m.eval()
    return m, params


def _engine_step_gpu(eng, m, params, x, ts, ctx, tc, r_out, seed=None, dev="cuda"):
    emb_all = m.conditioning_emb_all(ts.to(dev), 16, tc.to(dev))
    y = eng.forward_tape(x.to(dev), ts.to(dev), ctx.to(dev), 16, tc.to(dev), None, emb_all=emb_all, seed=seed)
    flat = torch.zeros(eng.lora_numel, device=dev)
    dx = eng.backward(r_out.to(dev), flat_grad=flat, accumulate=False)
    for p in params:
        p.grad = None
    emb_all.backward(eng.d_emb_all)
    grads, off = [], 0
    for p in params:
        gr = flat[off:off + p.numel()].view_as(p)
        grads.append((gr if p.grad is None else gr + p.grad).float().cpu())
        off += p.numel()
    return y.float().cpu(), dx.float().cpu(), grads


def test_tiny_student_on_device_vs_the_reference_lora_gradient_fixture():
    """unet_tiny_lora_grad.npz was computed by the REFERENCE (its UNetModel, its inject_trainable_lora_extended, autograd):
    the device engine is compared with it directly — output, d/d(latents), per-tensor norm and two random projections of all
    1150 gradients, the four rank-4 leaves in full."""
    from t2v_turbo_amd import lora
    from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
    from t2v_turbo_amd.native import HipOps
    from tests.golden.make_golden_lora_grad import SEED_R, digests, draw_lora
    g, gg = load("unet_tiny"), load("unet_tiny_lora_grad")
    m, params = _tiny_student(64, draw_lora)
    m = m.cuda()
    params = lora.lora_parameters(m)
    assert len(params) == 2 * int(gg["n_leaves"]) == 1150
    eng = UNetGradEngine(m, HipOps())
    eng.bind_lora(params)
    x, ts, ctx, tc = g["x"], g["ts"], g["ctx"], g["tc"]
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(SEED_R))
    for rep in range(2):
        y, dx, grads = _engine_step_gpu(eng, m, params, x, ts, ctx, tc, r_out)
        e_out, e_dx = rel_l2(y, gg["out"]), rel_l2(dx, gg["dx"])
        d, ref = torch.from_numpy(digests(grads)), gg["digests"]
        norm_err = ((d[:, 0] - ref[:, 0]).abs() / ref[:, 0])
        proj_err = ((d[:, 1:] - ref[:, 1:]).abs() / ref[:, :1])
        print(f"[fixture, pass {rep}] out {e_out:.3e} dx {e_dx:.3e}; norm err max {float(norm_err.max()):.3f} median "
              f"{float(norm_err.median()):.4f}; projection err / norm max {float(proj_err.max()):.3f} median "
              f"{float(proj_err.median()):.4f}", flush=True)
        assert e_out < OUT_TOL and e_dx < DX_TOL
        assert float(norm_err.max()) < 0.10, int(norm_err.argmax())
        assert float(proj_err.max()) < 0.30 and float(proj_err.median()) < 0.06, int(proj_err.max(dim=1).values.argmax())
        for k in ("g10", "g11", "g1148", "g1149"):
            assert rel_l2(grads[int(k[1:])], gg[k]) < 0.12, k


def test_mid_width_student_on_device_vs_the_reference_lora_gradient_fixture():
    """The same against tests/golden/unet_mid_lora_grad.npz (the reference run at model_channels = 128: 128-512-channel levels,
    2 / 4 / 8 heads): a second reference-generated anchor for the device gradient engine between the tiny fixture and the
    full-width gate, whose checker is this repository's own composite module."""
    from oracle.synth import manifest_of, synth_state_dict
    from t2v_turbo_amd import lora
    from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
    from t2v_turbo_amd.native import HipOps
    from t2v_turbo_amd.unet3d import UNetModel
    from tests.golden.make_golden_lora_grad import SEED_R, digests, draw_lora
    g, gg = load("unet_tiny"), load("unet_mid_lora_grad")
    m = UNetModel(**tiny_unet_params(model_channels=int(gg["width"]))).eval()
    m.load_state_dict(synth_state_dict(manifest_of(m)), strict=True)
    m.requires_grad_(False)
    lora.inject_trainable_lora_extended(m, r=64)
    draw_lora(lora.lora_parameters(m))
    m = m.eval().cuda()
    params = lora.lora_parameters(m)
    assert len(params) == 2 * int(gg["n_leaves"])
    eng = UNetGradEngine(m, HipOps())
    eng.bind_lora(params)
    x, ts, ctx, tc = g["x"], g["ts"], g["ctx"], g["tc"]
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(SEED_R))
    y, dx, grads = _engine_step_gpu(eng, m, params, x, ts, ctx, tc, r_out)
    e_out, e_dx = rel_l2(y, gg["out"]), rel_l2(dx, gg["dx"])
    d, ref = torch.from_numpy(digests(grads)), gg["digests"]
    norm_err = ((d[:, 0] - ref[:, 0]).abs() / ref[:, 0])
    proj_err = ((d[:, 1:] - ref[:, 1:]).abs() / ref[:, :1])
    print(f"[mid-width fixture] out {e_out:.3e} dx {e_dx:.3e}; norm err max {float(norm_err.max()):.3f} median {float(norm_err.median()):.4f}; "
          f"projection err / norm max {float(proj_err.max()):.3f} median {float(proj_err.median()):.4f}", flush=True)
    pe = proj_err.max(dim=1).values.double()
    qs = torch.quantile(pe, torch.tensor([0.5, 0.9, 0.99, 1.0], dtype=torch.float64))
    worst = torch.argsort(pe, descending=True)[:5].tolist()
    print("[mid-width fixture] per-tensor max projection error / norm: median {:.4f} p90 {:.4f} p99 {:.4f} max {:.4f}; worst tensors (index, err, ref norm): ".format(*qs.tolist())
          + ", ".join(f"({i}, {float(pe[i]):.3f}, {float(ref[i, 0]):.2e})" for i in worst), flush=True)
    assert e_out < OUT_TOL and e_dx < DX_TOL
    assert float(norm_err.max()) < 0.10, int(norm_err.argmax())
    # what the device achieves (round 5, profiles/r05_midwidth_projection_error.txt): per-tensor worst projection error / norm — median
    # 0.042, p90 0.082, p99 0.136, max 0.209 (a 1.3-norm tensor; the four next-worst 0.150-0.157).  A projection error is the
    # tensor's relative gradient error (0.04-0.06 with bf16 activations: cosine 0.998-0.999) times a standard-normal sample (two
    # random directions per tensor, tests/golden/make_golden_lora_grad.py::digests), so over ~2 300 samples the maximum sits at
    # ~3.5 sigma = 0.2 by construction: the bounds sit just above what was measured, not at the 0.30 the test started with
    assert float(pe.max()) < 0.25 and float(qs[2]) < 0.16 and float(qs[0]) < 0.06 and float(proj_err.median()) < 0.04, qs.tolist()


# ------------------------------------------------------------------------------------------------------------- (iv)
def test_train_mode_student_on_device_with_replayed_masks():
    """Train mode is what the step is timed in.  The device draws counter-based masks; the same masks (regenerated on the host
    from the recorded site geometry, bit-identical by tests/test_gpu_unet_grad.py::test_dropout_mask_is_the_emulated_one) are
    patched into the fp32 torch module on the CPU: output, d/d(latents) and every LoRA gradient must agree."""
    from t2v_turbo_amd.native import HipOps
    run_train_mode_with_replayed_masks("cuda", HipOps(), OUT_TOL, DX_TOL, 0.985, 0.12)


def test_train_mode_student_with_one_gemm_per_lora_leaf(monkeypatch):
    """The same gate with ``fuse_lora`` off (T2V_LORA_EPILOGUE=0): the up-projection and dropout of every LoRA leaf as launches of their
    own, the form that was the default until the epilogue form (now default: the test above runs it) was timed in round 4."""
    from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
    from t2v_turbo_amd.native import HipOps
    monkeypatch.setattr(UNetGradEngine, "fuse_lora", False)
    run_train_mode_with_replayed_masks("cuda", HipOps(), OUT_TOL, DX_TOL, 0.985, 0.12)


def run_train_mode_with_replayed_masks(dev, ops, out_tol, dx_tol, cos_min, ratio_tol):
    import copy
    from t2v_turbo_amd import lora
    from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
    from tests.mask_replay import SiteGeometrySpy, patch_engine_masks
    from tests.test_unet_lora_grad_cpu import _autograd

    def draw(params):
        gen = torch.Generator().manual_seed(7)
        with torch.no_grad():
            for p in params:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.05)

    g = load("unet_tiny")
    ref, rparams = _tiny_student(64, draw)
    m = copy.deepcopy(ref).to(dev)
    params = lora.lora_parameters(m)
    m.train(); ref.train()
    spy = SiteGeometrySpy(ops)
    eng = UNetGradEngine(m, ops)
    eng.bind_lora(params)
    mine = set(map(id, eng.engine_leaves()))
    for mod_d, mod_h in zip(m.modules(), ref.modules()):   # the conditioning branch is torch's in both runs: no random masks there
        if hasattr(mod_d, "lora_up") and id(mod_d) not in mine:
            mod_d.dropout.eval(); mod_h.dropout.eval()
    x, ts, ctx, tc = g["x"], g["ts"], g["ctx"], g["tc"]
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(5))
    seed = 0x5EED_1234_ABCD
    y, dx, grads = _engine_step_gpu(eng, m, params, x, ts, ctx, tc, r_out, seed=seed, dev=dev)       # recording pass
    y2, dx2, grads2 = _engine_step_gpu(eng, m, params, x, ts, ctx, tc, r_out, seed=seed, dev=dev)    # replayed lists, same seed
    assert torch.equal(y, y2) and rel_l2(dx2, dx) < 1e-6
    y3, _, _ = _engine_step_gpu(eng, m, params, x, ts, ctx, tc, r_out, seed=seed + 1, dev=dev)
    assert rel_l2(y3, y) > 1e-3, "a new seed must draw new masks"
    sites = eng.drop_sites
    assert len(sites) > 100 and set(spy.sites) == set(range(len(sites)))
    masks = spy.masks(seed)
    keep = torch.cat([masks[i].reshape(-1).float() for i in range(len(sites))]).mean()
    assert abs(float(keep) - 0.9) < 2e-3
    patch_engine_masks(ref, eng, masks)
    y_ref, dx_ref, g_ref = _autograd(ref, rparams, x, ts, ctx, 16, tc, None, r_out)
    e_out, e_dx = rel_l2(y, y_ref), rel_l2(dx, dx_ref)
    print(f"[train mode] out {e_out:.3e} dx {e_dx:.3e}", flush=True)
    names = {id(p): n for n, p in m.named_parameters()}
    rows = []
    for p, gq, r in zip(params, grads, g_ref):
        rn = float(r.double().norm())
        if rn == 0.0:
            assert float(gq.abs().max()) < 1e-6
            continue
        rows.append((names[id(p)], _cos(gq, r), float(gq.double().norm()) / rn, rn))
    _report("train mode, tiny", rows, cos_min, ratio_tol)
    assert e_out < out_tol and e_dx < dx_tol


# ------------------------------------------------------------------------------------------------------------- (iv')
def test_full_width_student_in_train_mode_with_replayed_masks():
    """The chain closed at the WIDTH and MODE that bench.py times ("what's weak" 1(i)): the full-width VC2 student
    (r = 64, 1 150 LoRA tensors) in TRAIN mode — LoRA epilogue with 16-bit masks, the split-K-aware choice of the LoRA form, the
    halo conv in the training forward, TemporalConvBlock dropouts — forward + backward on the device, against fp32 CPU autograd
    through the torch module with the ENGINE'S masks patched into every ``nn.Dropout`` it applied (tests/mask_replay.py; the masks
    are regenerated on the host from the recorded site geometry).  A 2-frame latent (1,4,2,40,64) bounds the host side of the default
    run (the mask tensors alone are ~1e9 elements at 16 frames): every level, width, tile choice and epilogue variant of the 16-frame
    step except the M of the launches; the 16-frame gate itself is an environment switch away and was run in round 6."""
    import bench
    from t2v_turbo_amd.native import HipOps
    # T2V_TEST_TRAIN_PARITY_FRAMES=16: the same gate at the timed 16 frames (3.3 minutes, most of it fp32 CPU autograd; round 6's run:
    # out 1.97e-2, dx 5.39e-2, cosine >= 0.9914, norm ratio within 2.6 % — profiles/r06_full_width_train_parity_16_frames.txt)
    frames = int(os.environ.get("T2V_TEST_TRAIN_PARITY_FRAMES", "2"))
    run_student_train_mode_vs_reference_oracle(torch.device("cuda", 0), HipOps(), bench.VC2_UNET, (1, 4, frames, 40, 64), 1150, 400,
                                               OUT_TOL, DX_TOL, 0.985, 0.12, seed_model=4321)


def run_student_train_mode_vs_reference_oracle(dev, ops, cfg, x_shape, n_lora, min_sites, out_tol, dx_tol, cos_min, ratio_tol, seed_model):
    """Body of the test above (the CPU suite dry-runs it at tiny width on the emulated backend: tests/test_train_parity_cpu.py)."""
    from oracle.lora_grad_oracle import student_reference
    from t2v_turbo_amd import lora
    from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
    from t2v_turbo_amd.unet3d import UNetModel
    from tests.mask_replay import SiteGeometrySpy, patch_engine_masks
    t0 = time.time()
    torch.manual_seed(seed_model)
    if dev.type == "cuda":
        torch.cuda.manual_seed_all(seed_model)
    with torch.device(dev):
        student = UNetModel(**cfg)
    g = torch.Generator(device=dev).manual_seed(seed_model)
    with torch.no_grad():
        for p in student.parameters():
            if float(p.abs().max()) == 0.0:
                p.normal_(0.0, 0.02, generator=g)
    student.requires_grad_(False)
    lora.inject_trainable_lora_extended(student, r=64)
    params = lora.lora_parameters(student)
    assert n_lora is None or len(params) == n_lora
    with torch.no_grad():
        for p in params:
            if float(p.abs().max()) == 0.0:
                p.normal_(0.0, 0.02, generator=g)
    student.train()
    spy = SiteGeometrySpy(ops)
    eng = UNetGradEngine(student, ops)
    eng.bind_lora(params)
    mine = set(map(id, eng.engine_leaves()))
    for mod in student.modules():   # the conditioning branch is torch's in both runs: no random masks there
        if hasattr(mod, "lora_up") and id(mod) not in mine:
            mod.dropout.eval()
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(*x_shape, generator=gen)
    ctx = torch.randn(x_shape[0], 77, cfg["context_dim"], generator=gen)
    tc = torch.randn(x_shape[0], cfg["time_cond_proj_dim"], generator=gen) if cfg.get("time_cond_proj_dim") else None
    r_out = torch.randn(x.shape, generator=gen)
    ts = torch.tensor([519] * x_shape[0])
    seed = 0x5EED_0005_ABCD
    y, dx, grads = _engine_step_gpu(eng, student, params, x, ts, ctx, tc, r_out, seed=seed, dev=dev)        # recording pass
    y2, dx2, grads2 = _engine_step_gpu(eng, student, params, x, ts, ctx, tc, r_out, seed=seed, dev=dev)     # replayed lists, same masks
    assert torch.isfinite(y).all() and torch.isfinite(dx).all()
    assert torch.equal(y, y2) and rel_l2(dx2, dx) < 1e-6
    sites = eng.drop_sites
    assert len(sites) > min_sites and set(spy.sites) == set(range(len(sites)))
    plan = next(reversed(eng.plans.values()))
    if "rec" in plan:
        print(f"device side done (+{time.time() - t0:.0f}s): {len(sites)} dropout sites, {len(plan['rec'])} forward / "
              f"{len(plan['rec_bwd'])} backward launches", flush=True)
    masks = spy.masks(seed)
    kept = sum(int(masks[i].sum()) for i in range(len(sites))) / sum(masks[i].numel() for i in range(len(sites)))
    assert abs(kept - (1.0 - 6553.0 / 65536.0)) < 2e-3, kept   # the 16-bit threshold's own keep rate
    print(f"host masks done (+{time.time() - t0:.0f}s): keep rate {kept:.5f}", flush=True)

    def prepare(ref):
        ref.train()
        leaves_dev = [mod for mod in student.modules() if hasattr(mod, "lora_up")]
        leaves_ref = [mod for mod in ref.modules() if hasattr(mod, "lora_up")]
        assert len(leaves_dev) == len(leaves_ref)
        for a, b in zip(leaves_dev, leaves_ref):
            if id(a) not in mine:
                b.dropout.eval()
        patch_engine_masks(ref, eng, masks)

    y_ref, dx_ref, g_ref = student_reference(student.state_dict(), cfg, 64, x, ts, ctx, 16, tc, r_out,
                                             threads=min(os.cpu_count() or 1, 64), checkpoint=False, prepare=prepare)
    print(f"host reference done (+{time.time() - t0:.0f}s)", flush=True)
    e_out, e_dx = rel_l2(y, y_ref), rel_l2(dx, dx_ref)
    print(f"[train mode, replayed masks, {tuple(x_shape)}] out rel-L2 {e_out:.3e}  d/d(latents) rel-L2 {e_dx:.3e}", flush=True)
    names = {id(p): n for n, p in student.named_parameters()}
    rows = []
    for p, gq, r in zip(params, grads, g_ref):
        rn = float(r.double().norm())
        if rn == 0.0:
            assert float(gq.abs().max()) < 1e-6
            continue
        rows.append((names[id(p)], _cos(gq, r), float(gq.double().norm()) / rn, rn))
    _report("train mode vs oracle", rows, cos_min, ratio_tol)
    assert e_out < out_tol and e_dx < dx_tol, (e_out, e_dx)


# ------------------------------------------------------------------------------------------------------------- (iii)
def test_trainer_route_two_distill_steps_with_an_optimizer_update_between():
    """What train_t2v_turbo_v1_lora.py does with the drop-in classes: `unet(noisy, t, **context)` under autocast ->
    `loss.backward()` -> clip -> optimizer step, twice.  The module routes itself to the gradient engine (`_auto_route` ->
    `_NativeStudent`); the frozen teacher to the inference engine.  Reference: the same two steps through the torch composite
    path in fp32 (ATen, same GPU), its parameters set to the device run's parameters before each step, so that step 2 checks
    that the engine's operand packs followed the FlatAdamW update (control: the reference with the OLD parameters is worse)."""
    run_trainer_route(torch.device("cuda", 0), None, 0.08, 0.97)


def run_trainer_route(dev, emu_factory, loss_tol, cos_min):
    """``emu_factory``: CPU dry run of the same body on the emulated op backend (tests/test_train_parity_cpu.py)."""
    import copy
    import warnings
    from t2v_turbo_amd import cd_math, lora
    from t2v_turbo_amd.dist import FlatGradSync
    from t2v_turbo_amd.distill import distill_step
    from t2v_turbo_amd.optim import FlatAdamW
    from t2v_turbo_amd.scheduler import T2VTurboScheduler
    from t2v_turbo_amd.unet3d import UNetModel
    from oracle.synth import synth_state_dict

    def draw(params):
        gen = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for p in params:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.05)

    on_gpu = emu_factory is None
    sd = synth_state_dict(manifest("unet_tiny"))
    teacher32 = UNetModel(**tiny_unet_params(time_cond_proj_dim=None)).eval()
    teacher32.load_state_dict({k: v for k, v in sd.items() if not k.startswith("time_cond_proj")}, strict=True)
    teacher32.requires_grad_(False)
    if on_gpu:
        teacher = copy.deepcopy(teacher32).to(dev, torch.bfloat16)
        teacher.dtype = torch.bfloat16
    else:
        teacher = copy.deepcopy(teacher32)
    teacher32 = teacher32.to(dev)
    teacher32.native_mode = "off"
    base, _ = _tiny_student(16, draw)
    student = copy.deepcopy(base).to(dev)           # device run: native_mode "auto"
    if not on_gpu:
        student._native_ops_factory = emu_factory
        student.native_mode = "train"
    ref = copy.deepcopy(base).to(dev)               # reference: torch composite, fp32
    ref.native_mode = "off"
    sched = T2VTurboScheduler()
    solver = cd_math.DDIMSolver(sched.alphas_cumprod.numpy(), ddim_timesteps=50).to(dev)
    gen = torch.Generator().manual_seed(0)
    lat = torch.randn(2, 4, 2, 8, 8, generator=gen).to(dev)
    pe, ue = torch.randn(2, 77, 128, generator=gen).to(dev), torch.randn(2, 77, 128, generator=gen).to(dev)
    rngs = [dict(index=torch.tensor([3, 40], device=dev), noise=torch.randn(lat.shape, generator=gen).to(dev), w=torch.tensor([6.0, 11.5])),
            dict(index=torch.tensor([25, 9], device=dev), noise=torch.randn(lat.shape, generator=gen).to(dev), w=torch.tensor([14.0, 5.5]))]
    params = lora.lora_parameters(student)
    sync = FlatGradSync(params)
    opt = FlatAdamW(params, sync, lr=2e-2, weight_decay=0.0)
    rparams = lora.lora_parameters(ref)
    rsync = FlatGradSync(rparams)
    kw = dict(loss_type="l2", max_grad_norm=1e9)

    def ref_step(rng, flat_param):
        with torch.no_grad():
            off = 0
            for p in rparams:
                p.copy_(flat_param[off:off + p.numel()].view_as(p)); off += p.numel()
        loss, _ = distill_step(ref, teacher32, solver, sched, lat, pe, ue, grad_sync=rsync, rng=rng, **kw)
        return float(loss.detach()), rsync.flat.clone()

    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)   # the ATen fallback announces itself with a RuntimeWarning: not here
        p0 = opt.flat_param.clone()
        loss1, _ = distill_step(student, teacher, solver, sched, lat, pe, ue, optimizer=opt, grad_sync=sync, rng=rngs[0],
                                autocast_dtype=torch.bfloat16 if on_gpu else None, **kw)
        g1 = sync.flat.clone()
        p1 = opt.flat_param.clone()
        loss2, _ = distill_step(student, teacher, solver, sched, lat, pe, ue, optimizer=opt, grad_sync=sync, rng=rngs[1],
                                autocast_dtype=torch.bfloat16 if on_gpu else None, **kw)
        g2 = sync.flat.clone()
    box = student._engine_box
    # eval-mode student (no dropout): the grad-mode forward lands on the gradient engine, the no-grad target forward on the
    # inference engine with the LoRA branch merged at pack time (train mode would take the gradient engine's forward-only twin)
    assert box.grad is not None and (box.enc is not None or box.engine is not None), "the student did not run on the native engines"
    assert not on_gpu or teacher._engine_box.engine is not None, "the teacher did not run on the inference engine"
    assert len(box.grad.plans) == 1 and box.grad._last["fwd_id"] >= 2, "step 2 must replay step 1's recorded plan"
    assert float((p1 - p0).abs().mean()) > 1e-2, "the optimizer did not move the parameters"
    r_loss1, r_g1 = ref_step(rngs[0], p0)
    r_loss2, r_g2 = ref_step(rngs[1], p1)
    _, r_g2_old = ref_step(rngs[1], p0)
    c1, c2, c2_old = _cos(g1, r_g1), _cos(g2, r_g2), _cos(g2, r_g2_old)
    print(f"[trainer route] loss {float(loss1):.5f} / {float(loss2):.5f} vs fp32 {r_loss1:.5f} / {r_loss2:.5f}; flat-gradient cosine "
          f"step 1 {c1:.4f}, step 2 {c2:.4f} (against the pre-update parameters: {c2_old:.4f})", flush=True)
    assert abs(float(loss1) - r_loss1) < loss_tol * abs(r_loss1) and abs(float(loss2) - r_loss2) < loss_tol * abs(r_loss2)
    assert c1 > cos_min and c2 > cos_min
    assert c2 - c2_old > 0.03, "step 2 looks like it ran on the parameters of step 1: operand packs not refreshed"


# ------------------------------------------------------------------------------------------------------------- a12
def test_native_checkpointing_on_device_is_the_tape_bit_for_bit():
    """``checkpoint_blocks`` (the reference's ``use_checkpoint``, lvdm/common.py:96-112 — yaml ``use_checkpoint: true``): each
    ResBlock / transformer keeps its input only and re-runs its forward inside the backward.  Train mode on the device: the
    recomputed dropout masks are the forward's, every kernel sees the same operands, so output, d/d(latents), d/d(emb_all) and all
    LoRA gradients equal the tape's BIT FOR BIT, from a smaller activation pool; replays of the recorded lists stay identical."""
    from t2v_turbo_amd import lora
    from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
    from t2v_turbo_amd.native import HipOps
    from tests.golden.make_golden_lora_grad import draw_lora
    g = load("unet_tiny")
    x, ts, ctx, tc = g["x"], g["ts"], g["ctx"], g["tc"]
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(5))
    res = {}
    for ck in (False, True):
        m, params = _tiny_student(64, draw_lora)
        m = m.cuda().train()
        params = lora.lora_parameters(m)
        eng = UNetGradEngine(m, HipOps())
        eng.checkpoint_blocks = ck
        eng.bind_lora(params)
        torch.manual_seed(3)                        # the conditioning branch's own dropouts (torch's generator)
        emb_all = m.conditioning_emb_all(ts.cuda(), 16, tc.cuda()).detach()
        outs = []
        for rep in range(3):                        # record, plain replay, replay
            y = eng.forward_tape(x.cuda(), ts.cuda(), ctx.cuda(), 16, tc.cuda(), None, emb_all=emb_all, seed=4242)
            flat = torch.zeros(eng.lora_numel, device="cuda")
            dx = eng.backward(r_out.cuda(), flat_grad=flat, accumulate=False)
            outs.append((y.clone(), dx.clone(), flat.clone(), eng.d_emb_all.clone()))
        for o in outs[1:]:
            assert all(torch.equal(a, b) for a, b in zip(o, outs[0]))
        plan = eng._last
        res[ck] = dict(out=outs[0], pool=eng.pool.bytes, n_fwd=len(plan["rec"]), n_bwd=len(plan["rec_bwd"]))
    a, b = res[False], res[True]
    print(f"[checkpoint] activation pool {a['pool'] / 2**20:.1f} MiB (tape) -> {b['pool'] / 2**20:.1f} MiB; launches forward "
          f"{a['n_fwd']} / {b['n_fwd']}, backward {a['n_bwd']} -> {b['n_bwd']}", flush=True)
    assert float(a["out"][2].abs().sum()) > 0
    for ta, tb in zip(a["out"], b["out"]):
        assert torch.equal(ta, tb)
    assert b["pool"] < 0.6 * a["pool"] and a["n_fwd"] == b["n_fwd"] and b["n_bwd"] > a["n_bwd"] + 0.8 * a["n_fwd"]


# ------------------------------------------------------------------------------------------------------------- (e)
def test_overlapped_gradient_exchange_over_rccl_one_rank():
    """The multi-rank exchange of the training step as the device runs it (train_t2v_turbo_v1_lora.py:1190 under DDP): the recorded
    backward list carries marker entries that all-reduce the gradient arena in segments over RCCL while later launches are still
    being issued; the gather into parameter order applies 1 / world; the conditioning branch's tensors go through
    ``FlatGradSync.all_reduce_mean`` as a subset.  One GPU per box, so: a one-rank ``nccl`` communicator with ``sync.force`` — the
    collectives really run on RCCL's stream between the engine's launches, and must hand over exactly the gradients of the plain
    path (a 1-rank sum is the identity: bit-identical), step after step (replays of the recorded list).
    (In a child interpreter: tests.util.run_isolated.)"""
    from tests.util import run_isolated
    run_isolated("tests.test_gpu_train_parity", "_body_overlapped_gradient_exchange_over_rccl_one_rank")


def _body_overlapped_gradient_exchange_over_rccl_one_rank():
    import torch.distributed as dist
    from t2v_turbo_amd import dist as tdist, lora
    from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
    from t2v_turbo_amd.native import HipOps
    from tests.golden.make_golden_lora_grad import draw_lora
    g = load("unet_tiny")
    x, ts, ctx, tc = (g[k].cuda() for k in ("x", "ts", "ctx", "tc"))
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(5)).cuda()
    m, _ = _tiny_student(64, draw_lora)
    m = m.cuda()
    params = lora.lora_parameters(m)
    sync = tdist.FlatGradSync(params)
    eng = UNetGradEngine(m, HipOps())
    eng.bind_lora(params)

    def step(use_sync):
        sync.zero_()
        emb_all = m.conditioning_emb_all(ts, 16, tc)
        eng.forward_tape(x, ts, ctx, 16, tc, None, emb_all=emb_all)
        dx = eng.backward(r_out, flat_grad=sync.flat, accumulate=True, grad_sync=sync if use_sync else None)
        emb_all.backward(eng.d_emb_all)
        rest = sync._rest_idx
        sync.all_reduce_mean()
        torch.cuda.synchronize()
        return dx.clone(), sync.flat.clone(), rest

    dx0, flat0, rest0 = step(False)
    assert rest0 is None and float(flat0.abs().sum()) > 0
    assert not dist.is_initialized()
    tdist.init_distributed("nccl", single_process_group=True)
    try:
        sync.force = True
        for rep in range(3):
            dx1, flat1, rest1 = step(True)
            assert len(eng._handles) == 8, len(eng._handles)
            assert rest1 is not None and rest1.numel() == sum(p.numel() for p in eng.conditioning_parameters()) > 0
            assert torch.equal(dx1, dx0) and torch.equal(flat1, flat0), rep
        markers = [e for e in eng._last["rec_bwd"] if e[2] == "allreduce_segment"]
        assert len(markers) == 8
        pos = [i for i, e in enumerate(eng._last["rec_bwd"]) if e[2] == "allreduce_segment"]
        print(f"[overlap] 8 segment markers at launch {pos} of {len(eng._last['rec_bwd'])}", flush=True)
        assert pos[0] < 0.5 * len(eng._last["rec_bwd"])     # the first piece leaves long before the backward ends
        # hipGraph replay: the backward list is cut at the markers — one graph per run of launches, the all-reduces between them
        eng.use_graph = True
        for rep in range(3):
            dx2, flat2, rest2 = step(True)
            assert len(eng._handles) == 8 and rest2 is not None
            assert torch.equal(dx2, dx0) and torch.equal(flat2, flat0), rep
        built = eng._last["graph_rec_bwd"]
        assert sum(1 for g, h in built if g is None) == 8 and sum(1 for g, h in built if g is not None) >= 8
        assert len(eng._last["graph_rec"]) == 1 and "graph_failed" not in eng._last
        eng.use_graph = False
        torch.cuda.synchronize()
        print("BODY_OK", flush=True)
    finally:
        sync.force = False
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------- (v) full fine-tuning
def test_full_fine_tuning_on_device_vs_the_reference_gradient_fixture():
    """Row a20 (train_latent_t2v_turbo_v2.py:669,798-816,1262: every UNet parameter trainable, no LoRA) on MI355X: the module route lands on
    the native gradient engine with base-weight gradients (engine_full.py: t2v_wgrad_tn / t2v_im2col_bf16 / t2v_norm_affine_grad), WITHOUT
    the torch-composite warning, and one step reproduces the imported reference's own gradients of all 1485 parameters
    (tests/golden/unet_tiny_full_grad.npz: output, d/d latents, per-parameter norm + two random projections, ten small parameters in full).
    Then an optimizer-style update of every weight: the SAME recorded plan (packs re-filled in place) must give the gradients of the new
    weights — checked against fp32 CPU autograd through the torch module."""
    import copy
    import warnings
    from oracle.synth import synth_state_dict
    from t2v_turbo_amd.unet3d import UNetModel
    from tests.golden.make_golden_full_grad import SEED_R
    from tests.test_unet_full_grad_cpu import _fixture_step, check_against_reference_fixture
    g, gg = load("unet_tiny"), load("unet_tiny_full_grad")
    ref = UNetModel(**tiny_unet_params())
    ref.load_state_dict(synth_state_dict(manifest("unet_tiny")), strict=True)
    ref.requires_grad_(True)
    ref.eval()
    m = copy.deepcopy(ref).cuda()
    names = [n for n, _ in m.named_parameters()]
    r_out = torch.randn(g["x"].shape, generator=torch.Generator().manual_seed(SEED_R))
    dev = lambda t: t.cuda()   # noqa: E731
    args = (dev(g["x"]), dev(g["ts"]), dev(g["ctx"]), dev(g["tc"]), dev(r_out))
    for rep in range(2):   # recording pass, then a replay
        with warnings.catch_warnings():
            warnings.simplefilter("error")     # the ATen-route warning must not fire
            y, dx, grads = _fixture_step(m, *args, "auto")
        assert m._engine_box.full is not None and len(m._engine_box.full.plans) == 1
        check_against_reference_fixture(y.cpu(), dx.cpu(), [t.cpu() for t in grads], names, gg, OUT_TOL, DX_TOL, 0.10, (0.30, 0.06), 0.12)
    # every weight moves (an optimizer step), three times: same plan, new packs — the first refresh runs eagerly, the second is captured
    # as one hipGraph, the third replays it (engine_full._refresh)
    plan = next(iter(m._engine_box.full.plans.values()))
    gen = torch.Generator().manual_seed(5)
    y_prev = gg["out"]
    for upd in range(3):
        with torch.no_grad():
            for p, q in zip(ref.parameters(), m.parameters()):
                d = torch.randn(p.shape, generator=gen) * 0.03 * float(p.abs().mean() + 1e-3)
                p.add_(d)
                q.add_(d.cuda())
        y_r, dx_r, g_r = _fixture_step(ref, g["x"], g["ts"], g["ctx"], g["tc"], r_out, "off")
        y, dx, grads = _fixture_step(m, *args, "auto")
        assert next(iter(m._engine_box.full.plans.values())) is plan
        assert rel_l2(y_r, y_prev) > 1e-3, "the update must change the output for this check to mean anything"
        y_prev = y_r
        assert rel_l2(y.cpu(), y_r) < OUT_TOL and rel_l2(dx.cpu(), dx_r) < DX_TOL, upd
        cos = torch.tensor([float(torch.nn.functional.cosine_similarity(a.cpu().double().reshape(1, -1), b.double().reshape(1, -1)))
                            for a, b in zip(grads, g_r) if float(b.abs().max()) > 0])
        print(f"[full fine-tuning, after update {upd + 1}] gradient cosine min {float(cos.min()):.4f} median {float(cos.median()):.4f}", flush=True)
        assert float(cos.min()) > 0.97 and float(cos.median()) > 0.995, upd
    eng = m._engine_box.full
    if eng.refresh_graph:
        assert eng._refresh_state["graph"] is not None and not eng._refresh_state["failed"], "the pack refresh was not captured"


def test_full_fine_tuning_mid_width_on_device_vs_the_reference_gradient_fixture():
    """The same against tests/golden/unet_mid_full_grad.npz (the reference run at model_channels = 128: 128-512-channel levels, 2 / 4 / 8
    heads, 228 M parameters): a second reference-generated anchor for the base-weight gradients between the tiny fixture and the full width."""
    import warnings
    from oracle.synth import manifest_of, synth_state_dict
    from t2v_turbo_amd.unet3d import UNetModel
    from tests.golden.make_golden_full_grad import SEED_R
    from tests.test_unet_full_grad_cpu import _fixture_step, check_against_reference_fixture
    g, gg = load("unet_tiny"), load("unet_mid_full_grad")
    m = UNetModel(**tiny_unet_params(model_channels=int(gg["width"])))
    m.load_state_dict(synth_state_dict(manifest_of(m)), strict=True)
    m = m.requires_grad_(True).eval().cuda()
    names = [n for n, _ in m.named_parameters()]
    r_out = torch.randn(g["x"].shape, generator=torch.Generator().manual_seed(SEED_R))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        y, dx, grads = _fixture_step(m, g["x"].cuda(), g["ts"].cuda(), g["ctx"].cuda(), g["tc"].cuda(), r_out.cuda(), "auto")
    assert m._engine_box.full is not None
    check_against_reference_fixture(y.cpu(), dx.cpu(), [t.cpu() for t in grads], names, gg, OUT_TOL, DX_TOL, 0.10, (0.30, 0.06), 0.12)


def test_full_fine_tuning_train_mode_runs_with_live_temporal_dropouts():
    """The v2 student is in train mode (:669): the TemporalConvBlock dropouts are live (counter-based masks, the same sites as LoRA
    training) — the route stays native, outputs and gradients are finite, and two calls draw different masks."""
    import warnings
    from oracle.synth import synth_state_dict
    from t2v_turbo_amd.unet3d import UNetModel
    g = load("unet_tiny")
    m = UNetModel(**tiny_unet_params())
    m.load_state_dict(synth_state_dict(manifest("unet_tiny")), strict=True)
    m = m.cuda().requires_grad_(True).train()
    outs = []
    for _ in range(2):
        for p in m.parameters():
            p.grad = None
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            y = m(g["x"].cuda(), g["ts"].cuda(), context=g["ctx"].cuda(), fps=16, timestep_cond=g["tc"].cuda())
        y.float().pow(2).mean().backward()
        assert torch.isfinite(y).all() and all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
        outs.append(y.detach().float().cpu())
    assert m._engine_box.full is not None and rel_l2(outs[0], outs[1]) > 1e-4


def test_full_fine_tuning_at_full_width_vs_cpu_autograd():
    """Row a20 at the WIDTH the reference trains at: the VideoCrafter2 UNet (1 413 M parameters), every parameter trainable, eval mode, a
    2-frame latent (1,4,2,40,64) — every level, width and leaf kind of the 16-frame step (320 / 640 / 1 280-channel convs with their
    im2col matrices, the 2 560-channel concat GroupNorms, the 10 240-column GEGLU pre-activation, weight-gradient products larger than the split-K workspace);
    forward + backward through the module route on the device against fp32 CPU autograd through the same module (checkpointed, as the
    LoRA gate's oracle): output, d/d(latents) and the gradient of EVERY parameter by cosine and norm.  T2V_TEST_TRAIN_PARITY_FRAMES
    raises the frame count."""
    import copy
    import warnings
    import bench
    from t2v_turbo_amd.unet3d import UNetModel
    frames = int(os.environ.get("T2V_TEST_TRAIN_PARITY_FRAMES", "2"))
    t0 = time.time()
    dev = torch.device("cuda", 0)
    m = bench.build_model(dev, torch.float32)
    m.requires_grad_(True)
    m.eval()
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, frames, 40, 64, generator=gen)
    ctx = torch.randn(1, 77, bench.VC2_UNET["context_dim"], generator=gen)
    _, _, tc = bench.synth_inputs(dev, torch.float32)
    ts = torch.tensor([999])
    r_out = torch.randn(x.shape, generator=gen)
    with warnings.catch_warnings():
        warnings.simplefilter("error")      # the torch-composite route warns: it must not be taken
        xg = x.to(dev).requires_grad_(True)
        y = m(xg, ts.to(dev), context=ctx.to(dev), fps=16, timestep_cond=tc)
        (y * r_out.to(dev)).sum().backward()
    assert m._engine_box.full is not None
    torch.cuda.synchronize()
    got = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()}
    y_d, dx_d = y.detach().float().cpu(), xg.grad.detach().float().cpu()
    print(f"device step done (+{time.time() - t0:.0f}s)", flush=True)
    # fp32 CPU autograd through the same module, checkpointed
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    sd = {k: v.detach().to("cpu", torch.float32) for k, v in m.state_dict().items()}
    with torch.device("meta"):
        ref = UNetModel(**dict(bench.VC2_UNET, use_checkpoint=True))
    ref = ref.to_empty(device="cpu")
    ref.load_state_dict(sd, strict=True)
    ref.requires_grad_(True)
    ref.eval()
    ref.native_mode = "off"
    xr = x.clone().requires_grad_(True)
    y_r = ref(xr, ts, context=ctx, fps=16, timestep_cond=tc.detach().float().cpu())
    (y_r * r_out).sum().backward()
    print(f"host reference done (+{time.time() - t0:.0f}s)", flush=True)
    e_out, e_dx = rel_l2(y_d, y_r.detach()), rel_l2(dx_d, xr.grad)
    rows = []
    for n, p in ref.named_parameters():
        r, g = p.grad, got[n]
        rn = float(r.double().norm())
        if rn == 0.0:
            assert float(g.abs().max()) < 1e-6, n
            continue
        rows.append((n, _cos(g, r), float(g.double().norm()) / rn, rn))
    print(f"[full fine-tuning, full width, {frames} frames] out rel-L2 {e_out:.3e}  d/d(latents) rel-L2 {e_dx:.3e}", flush=True)
    _report("full fine-tuning, full width", rows, 0.98, 0.12)
    assert e_out < OUT_TOL and e_dx < DX_TOL
