"""LoRA weight gradients of the student UNet on the native engine's dataflow (CPU, emulated op backend, fp32): every
``lora_up`` / ``lora_down`` gradient, d(loss)/d(latents) and d(loss)/d(emb_all) must match torch autograd through the
LoRA-injected, reference-shaped module (utils/lora.py:45-50,124-129,204-209 forward; the student's backward of
train_t2v_turbo_v1_lora.py:1190).  Pins: the un-merged LoRA branch (t = x*D, z = s t U^T as the base leaf's residual), the
rank-r weight-gradient GEMMs (token-contracted, transposed operands, zero padding to rank / K 64), the gathered rank-r
gradient for conv leaves (stride 1 / stride 2 / nearest-x2 / temporal), the GEGLU row permutation, grouped q/k/v leaves, the
per-layer text K/V with their dK/dV, the 4-channel entry / exit convs, and the index maps between parameter layout, operand
packs and gradient arena."""
import torch

from oracle.synth import synth_state_dict
from t2v_turbo_amd import lora
from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
from t2v_turbo_amd.unet3d import UNetModel
from tests.emu_ops import EmuOps
from tests.util import load, manifest, rel_l2, tiny_unet_params


def _student(fixture, rank, **cfg):
    m = UNetModel(**tiny_unet_params(**cfg)).eval()
    m.load_state_dict(synth_state_dict(manifest(fixture)), strict=True)
    m.requires_grad_(False)
    lora.inject_trainable_lora_extended(m, r=rank)
    params = lora.lora_parameters(m)
    gen = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for p in params:  # lora_up is zero-initialised: draw both factors, or every lora_down gradient is zero
            p.copy_(torch.randn(p.shape, generator=gen) * 0.05)
    m.eval()
    for i, mod in enumerate(mm for mm in m.modules() if hasattr(mm, "lora_up")):
        mod.scale = 1.0 if i % 3 else 0.5  # the scale is per leaf
    return m, params


def _autograd(m, params, x, ts, ctx, fps, tc, mc, r_out):
    xg = x.clone().requires_grad_(True)
    m.native_mode = "off"
    for p in params:
        p.grad = None
    kw = {} if mc is None else {"motion_cond": mc}
    y = m(xg, ts, context=ctx, fps=fps, timestep_cond=tc, **kw)
    (y * r_out).sum().backward()
    return y.detach(), xg.grad, [p.grad.clone() for p in params]


def _engine_step(eng, m, params, x, ts, ctx, fps, tc, mc, r_out):
    emb_all = m.conditioning_emb_all(ts, fps, tc, mc)
    y = eng.forward_tape(x, ts, ctx, fps, tc, mc, emb_all=emb_all)
    flat = torch.zeros(eng.lora_numel)
    dx = eng.backward(r_out, flat_grad=flat, accumulate=False)
    for p in params:
        p.grad = None
    emb_all.backward(eng.d_emb_all)  # the M = B-row conditioning branch stays with torch
    grads, off = [], 0
    for p in params:
        g = flat[off:off + p.numel()].view_as(p)
        grads.append(g if p.grad is None else g + p.grad)
        off += p.numel()
    return y, dx, grads


def _compare(params, got, ref, m, tol=2e-4, max_zero=0):
    names = {id(p): n for n, p in m.named_parameters()}
    worst, zeros = (0.0, None), 0
    for p, g, r in zip(params, got, ref):
        if float(r.abs().max()) == 0:  # e.g. q / k of a one-token spatial attention: the softmax is constant
            assert float(g.abs().max()) < 1e-7, names[id(p)]
            zeros += 1
            continue
        e = rel_l2(g, r)
        if e > worst[0]:
            worst = (e, names[id(p)])
    assert worst[0] < tol, worst
    assert zeros <= max_zero, zeros


def test_lora_gradients_match_autograd():
    g = load("unet_tiny")
    m, params = _student("unet_tiny", 64)
    x, ts, ctx, tc = g["x"], g["ts"], g["ctx"], g["tc"]
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(5))
    eng = UNetGradEngine(m, EmuOps(strict=True))
    eng.bind_lora(params)
    # every injected leaf is either the engine's or the conditioning branch's
    assert len(eng.engine_leaves()) * 2 + len(eng.conditioning_parameters()) == len(params)
    y_ref, dx_ref, g_ref = _autograd(m, params, x, ts, ctx, 16, tc, None, r_out)
    y, dx, grads = _engine_step(eng, m, params, x, ts, ctx, 16, tc, None, r_out)
    assert rel_l2(y, y_ref) < 2e-5
    assert rel_l2(dx, dx_ref) < 1e-4
    _compare(params, grads, g_ref, m, max_zero=0)
    # an optimizer step later: the plan is replayed, the operand packs follow the parameters
    with torch.no_grad():
        gen = torch.Generator().manual_seed(9)
        for p in params:
            p.add_(torch.randn(p.shape, generator=gen) * 0.01)
    x2 = torch.randn(x.shape, generator=torch.Generator().manual_seed(3))
    n_plans = len(eng.plans)
    y_ref, dx_ref, g_ref = _autograd(m, params, x2, torch.tensor([519]), ctx, 24, tc, None, r_out)
    y, dx, grads = _engine_step(eng, m, params, x2, torch.tensor([519]), ctx, 24, tc, None, r_out)
    assert len(eng.plans) == n_plans, "a LoRA update must not invalidate the recorded plan"
    assert rel_l2(y, y_ref) < 2e-5
    assert rel_l2(dx, dx_ref) < 1e-4
    _compare(params, grads, g_ref, m, max_zero=0)


def test_lora_gradients_rank16_two_clips_motion_cond():
    """rank < 64 (operands zero-padded to the K granularity), two clips (per-clip text K/V and column sums), motion cond."""
    g = load("unet_tiny_mg_b2")
    m, params = _student("unet_tiny_mg_b2", 16, motion_cond_proj_dim=256)
    x, ts, ctx, tc, mc = g["x"], g["ts"], g["ctx"], g["tc"], g["mc"]
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(11))
    eng = UNetGradEngine(m, EmuOps(strict=True))
    eng.bind_lora(params)
    y_ref, dx_ref, g_ref = _autograd(m, params, x, ts, ctx, 8, tc, mc, r_out)
    y, dx, grads = _engine_step(eng, m, params, x, ts, ctx, 8, tc, mc, r_out)
    assert rel_l2(y, y_ref) < 2e-5
    assert rel_l2(dx, dx_ref) < 1e-4
    _compare(params, grads, g_ref, m, max_zero=8)


def test_lora_gradients_match_the_reference_fixture():
    """tests/golden/unet_tiny_lora_grad.npz: autograd through the REFERENCE UNetModel with the reference's own
    ``inject_trainable_lora_extended`` (tests/golden/make_golden_lora_grad.py).  Our injection order / shapes, the
    module's torch path and the native engine must all reproduce it."""
    from tests.golden.make_golden_lora_grad import SEED_R, digests, draw_lora
    g, gg = load("unet_tiny"), load("unet_tiny_lora_grad")
    m = UNetModel(**tiny_unet_params()).eval()
    m.load_state_dict(synth_state_dict(manifest("unet_tiny")), strict=True)
    m.requires_grad_(False)
    lora.inject_trainable_lora_extended(m, r=64)
    params = lora.lora_parameters(m)
    assert len(params) == 2 * int(gg["n_leaves"]) == 1150
    assert [list(p.shape) + [0] * (5 - p.dim()) for p in params] == gg["shapes"].tolist()  # same leaves, same order
    draw_lora(params)
    m.eval()
    x, ts, ctx, tc = g["x"], g["ts"], g["ctx"], g["tc"]
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(SEED_R))

    def check(y, dx, grads, tol):
        assert rel_l2(y, gg["out"]) < tol / 5
        assert rel_l2(dx, gg["dx"]) < tol
        d, ref = torch.from_numpy(digests(grads)), gg["digests"]
        assert ref.shape == (1150, 3)
        # norms to relative tolerance; projections against the tensor's norm (they can be near zero themselves)
        assert float(((d[:, 0] - ref[:, 0]).abs() / ref[:, 0]).max()) < tol
        assert float(((d[:, 1:] - ref[:, 1:]).abs() / ref[:, :1]).max()) < tol * 30
        for k in ("g10", "g11", "g1148", "g1149"):  # the rank-4 entry / exit conv leaves in full
            assert rel_l2(grads[int(k[1:])], gg[k]) < tol

    y_ref, dx_ref, g_ref = _autograd(m, params, x, ts, ctx, 16, tc, None, r_out)
    check(y_ref, dx_ref, g_ref, 1e-4)
    eng = UNetGradEngine(m, EmuOps(strict=True))
    eng.bind_lora(params)
    y, dx, grads = _engine_step(eng, m, params, x, ts, ctx, 16, tc, None, r_out)
    check(y, dx, grads, 2e-4)


def test_module_dispatch_native_train_mode():
    """``unet.native_mode = "train"``: the module's own ``forward`` / ``loss.backward()`` (what the reference's trainer
    calls, train_t2v_turbo_v1_lora.py:1022-1028,1190) run on the gradient engine through an autograd.Function; a no-grad
    forward between the student's forward and its backward (the target forward, :1163-1170) does not disturb the tape."""
    g = load("unet_tiny")
    m, params = _student("unet_tiny", 64)
    x, ts, ctx, tc = g["x"], g["ts"], g["ctx"], g["tc"]
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(5))
    y_ref, dx_ref, g_ref = _autograd(m, params, x, ts, ctx, 16, tc, None, r_out)
    with torch.no_grad():
        y2_ref = m(x * 0.5, torch.tensor([279]), context=ctx, fps=16, timestep_cond=tc)
    m._native_ops_factory = lambda: EmuOps(strict=True)
    m.native_mode = "train"
    for p in params:
        p.grad = None
    xg = x.clone().requires_grad_(True)
    y = m(xg, ts, context=ctx, fps=16, timestep_cond=tc)
    with torch.no_grad():
        y2 = m(x * 0.5, torch.tensor([279]), context=ctx, fps=16, timestep_cond=tc)
    (y * r_out).sum().backward()
    assert rel_l2(y.detach(), y_ref) < 2e-5 and rel_l2(y2, y2_ref) < 2e-5
    assert rel_l2(xg.grad, dx_ref) < 1e-4
    _compare(params, [p.grad for p in params], g_ref, m)
    # a base weight that requires grad is refused (frozen packs)
    m.out[2].conv.weight.requires_grad_(True)
    try:
        m(x, ts, context=ctx, fps=16, timestep_cond=tc)
    except RuntimeError:
        pass
    else:
        raise AssertionError


def test_train_mode_dropout_masks_and_gradients():
    """Train-mode student (train_t2v_turbo_v1_lora.py:641): Dropout(0.1) on every LoRA branch and in the temporal conv blocks.
    The engine draws counter-based masks (t2v_dropout_bf16) and regenerates them in the backward; torch's own random stream
    cannot be matched, so the masks the engine used are replayed inside the torch module (re-laid into torch's row orders)
    and everything — output, d/d(latents), all LoRA gradients — must agree with autograd."""
    g = load("unet_tiny")
    m, params = _student("unet_tiny", 64)
    m.train()
    eng = UNetGradEngine(m, EmuOps(strict=True))
    eng.bind_lora(params)
    for mod in m.modules():  # the conditioning branch is torch's in both runs: keep its random masks out of the comparison
        if hasattr(mod, "lora_up") and mod not in eng.engine_leaves():
            mod.dropout.eval()
    x, ts, ctx, tc = g["x"], g["ts"], g["ctx"], g["tc"]
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(5))
    eng.ops.masks = {}
    emb_all = m.conditioning_emb_all(ts, 16, tc, None)
    y = eng.forward_tape(x, ts, ctx, 16, tc, None, emb_all=emb_all, seed=1234)
    flat = torch.zeros(eng.lora_numel)
    dx = eng.backward(r_out, flat_grad=flat, accumulate=False)
    for p in params:
        p.grad = None
    emb_all.backward(eng.d_emb_all)
    grads, off = [], 0
    for p in params:
        gr = flat[off:off + p.numel()].view_as(p)
        grads.append(gr if p.grad is None else gr + p.grad)
        off += p.numel()
    sites = eng.drop_sites
    n_lora = sum(1 for _, kind, _ in sites if kind != "tconv")
    assert n_lora > 100 and len(sites) - n_lora > 20
    keep_frac = torch.cat([eng.ops.masks[i].reshape(-1).float() for i in range(len(sites))]).mean()
    assert abs(float(keep_frac) - 0.9) < 2e-3
    # ---- replay the masks in the torch module (tests/mask_replay.py, shared with the GPU suite) --------------------------
    from tests.mask_replay import patch_engine_masks
    patch_engine_masks(m, eng, eng.ops.masks)
    y_ref, dx_ref, g_ref = _autograd(m, params, x, ts, ctx, 16, tc, None, r_out)
    assert rel_l2(y, y_ref) < 2e-5
    assert rel_l2(dx, dx_ref) < 1e-4
    _compare(params, grads, g_ref, m)
    # a different seed gives a different output; the same seed the same one (the backward relies on that)
    y2 = eng.forward_tape(x, ts, ctx, 16, tc, None, emb_all=emb_all.detach(), seed=99)
    y3 = eng.forward_tape(x, ts, ctx, 16, tc, None, emb_all=emb_all.detach(), seed=1234)
    assert rel_l2(y2, y) > 1e-3 and torch.equal(y3, y)


def test_record_replay_protocol_of_the_native_backend(monkeypatch):
    """The engine on the native backend's record / replay protocol (tests.emu_ops.ReplayOps): launch lists recorded once,
    every later step = refresh static inputs + LoRA operand packs + seed, then re-issue the lists.  Three steps with different
    inputs and updated LoRA tensors against autograd; the third in train mode is a new plan (dropout sites) and must still
    run."""
    from tests.emu_ops import ReplayOps
    g = load("unet_tiny")
    m, params = _student("unet_tiny", 64)
    ops = ReplayOps()
    eng = UNetGradEngine(m, ops)
    eng.bind_lora(params)
    x, ts, ctx, tc = g["x"], g["ts"], g["ctx"], g["tc"]
    gen = torch.Generator().manual_seed(5)
    for step in range(3):
        r_out = torch.randn(x.shape, generator=gen)
        xs = x if step == 0 else torch.randn(x.shape, generator=gen)
        tss = ts if step == 0 else torch.tensor([279 + 240 * step])
        y_ref, dx_ref, g_ref = _autograd(m, params, xs, tss, ctx, 16 + step, tc, None, r_out)
        m.native_mode = "auto"
        y, dx, grads = _engine_step(eng, m, params, xs, tss, ctx, 16 + step, tc, None, r_out)
        assert rel_l2(y, y_ref) < 2e-5, step
        assert rel_l2(dx, dx_ref) < 1e-4, step
        _compare(params, grads, g_ref, m)
        with torch.no_grad():
            for p in params:
                p.add_(torch.randn(p.shape, generator=gen) * 0.01)
    assert len(eng.plans) == 1 and ops.replays >= 5  # step 0: record fwd + bwd, replay fwd, replay bwd; then two lists per step
    # forward-only use in between (the distillation step's target forward) leaves the next backward intact
    emb_all = m.conditioning_emb_all(ts, 16, tc, None)
    eng.forward_tape(x * 0.3, ts, ctx, 16, tc, None, emb_all=emb_all.detach())
    r_out = torch.randn(x.shape, generator=gen)
    y_ref, dx_ref, g_ref = _autograd(m, params, x, ts, ctx, 16, tc, None, r_out)
    y, dx, grads = _engine_step(eng, m, params, x, ts, ctx, 16, tc, None, r_out)
    assert rel_l2(dx, dx_ref) < 1e-4
    _compare(params, grads, g_ref, m)
    # train mode: a new plan (dropout sites recorded into the lists, seed read from its static buffer at replay time); it REPLACES
    # the eval-mode one — the LoRA operand / gradient arenas are per engine, and a kept plan would point into freed arenas
    m.train()
    emb_all = m.conditioning_emb_all(ts, 16, tc, None)
    y1 = eng.forward_tape(x, ts, ctx, 16, tc, None, emb_all=emb_all.detach(), seed=11)
    y2 = eng.forward_tape(x, ts, ctx, 16, tc, None, emb_all=emb_all.detach(), seed=12)
    y3 = eng.forward_tape(x, ts, ctx, 16, tc, None, emb_all=emb_all.detach(), seed=11)
    assert len(eng.plans) == 1 and torch.equal(y1, y3) and rel_l2(y2, y1) > 1e-3
    assert torch.isfinite(eng.backward(r_out, flat_grad=torch.zeros(eng.lora_numel))).all()


def test_flash_attention_backward_variant_of_the_engine():
    """``flash_attn_bwd``: the spatial self-attention backward as ONE op (csrc/attention_bwd.hip) instead of batched GEMMs around
    materialised probabilities — same gradients, in LoRA training and with frozen weights (V^T-from-GEMM layout)."""
    g = load("unet_tiny")
    m, params = _student("unet_tiny", 64)
    x, ts, ctx, tc = g["x"], g["ts"], g["ctx"], g["tc"]
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(5))
    y_ref, dx_ref, g_ref = _autograd(m, params, x, ts, ctx, 16, tc, None, r_out)
    eng = UNetGradEngine(m, EmuOps(strict=True))
    eng.flash_attn_bwd = True
    eng.bind_lora(params)
    y, dx, grads = _engine_step(eng, m, params, x, ts, ctx, 16, tc, None, r_out)
    assert eng.ops.calls.count("attn_spatial_bwd") == 16 and eng.ops.calls.count("softmax_bwd_rows") == 16  # (text cross-attention keeps the GEMM form)
    assert rel_l2(y, y_ref) < 2e-5 and rel_l2(dx, dx_ref) < 1e-4
    _compare(params, grads, g_ref, m)
    # frozen weights (data gradient only): V comes as V^T out of a GEMM, the kernel reads its per-head transpose
    m.requires_grad_(False)
    eng2 = UNetGradEngine(m, EmuOps(strict=True))
    eng2.flash_attn_bwd = True
    y2 = eng2.forward_tape(x, ts, ctx, 16, tc, None)
    assert rel_l2(y2, y_ref) < 2e-5 and rel_l2(eng2.backward(r_out), dx_ref) < 1e-4


def test_tn_weight_gradient_variant_of_the_engine():
    """``tn_wgrad``: every LoRA weight gradient (and the per-clip column sums) by the token-contracted op on the token-major
    operands — no transposed copies — with and without the train-mode dropout; same gradients."""
    g = load("unet_tiny_mg_b2")
    m, params = _student("unet_tiny_mg_b2", 16, motion_cond_proj_dim=256)
    x, ts, ctx, tc, mc = g["x"], g["ts"], g["ctx"], g["tc"], g["mc"]
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(11))
    y_ref, dx_ref, g_ref = _autograd(m, params, x, ts, ctx, 8, tc, mc, r_out)
    eng = UNetGradEngine(m, EmuOps(strict=True))
    eng.tn_wgrad = True
    eng.flash_attn_bwd = True
    eng.bind_lora(params)
    y, dx, grads = _engine_step(eng, m, params, x, ts, ctx, 8, tc, mc, r_out)
    calls = eng.ops.calls
    # one launch pair per LoRA group (dU of its leaves + per-clip column sums + dD of its input parts)
    assert calls.count("wgrad_tn_group") > 400 and calls.count("wgrad_tn") == 0
    assert calls.count("transpose_pad") < 200   # (what is left: the attention backward's operands)
    assert rel_l2(y, y_ref) < 2e-5 and rel_l2(dx, dx_ref) < 1e-4
    _compare(params, grads, g_ref, m, max_zero=8)
    # ``group_wgrad = False``: one t2v_wgrad_tn pair per product — the same numbers
    eng1 = UNetGradEngine(m, EmuOps(strict=True))
    eng1.tn_wgrad, eng1.flash_attn_bwd, eng1.group_wgrad = True, True, False
    eng1.bind_lora(params)
    y1, dx1, grads1 = _engine_step(eng1, m, params, x, ts, ctx, 8, tc, mc, r_out)
    assert eng1.ops.calls.count("wgrad_tn") > 1000 and eng1.ops.calls.count("wgrad_tn_group") == 0
    assert torch.equal(y1, y) and torch.equal(dx1, dx) and all(torch.equal(a, b) for a, b in zip(grads1, grads))


def test_native_checkpointing_recomputes_the_same_function():
    """``checkpoint_blocks`` (the reference's ``use_checkpoint``, lvdm/common.py:96-112): every ResBlock / transformer keeps only
    its input and re-runs its forward inside the backward.  Train mode (dropout on): the recomputed masks must be the forward's
    own — output, d/d(latents) and every LoRA gradient are bit-identical to the tape, with a smaller activation pool and exactly
    one more forward's worth of launches in the backward."""
    g = load("unet_tiny")
    x, ts, ctx, tc = g["x"], g["ts"], g["ctx"], g["tc"]
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(5))
    res = {}
    for ck in (False, True):
        m, params = _student("unet_tiny", 64)
        m.train()
        eng = UNetGradEngine(m, EmuOps(strict=True))
        eng.checkpoint_blocks = ck
        eng.bind_lora(params)
        torch.manual_seed(3)                        # the conditioning branch's dropouts are torch's: same draw in both runs
        emb_all = m.conditioning_emb_all(ts, 16, tc, None).detach()
        y = eng.forward_tape(x, ts, ctx, 16, tc, None, emb_all=emb_all, seed=77)
        n_fwd = len(eng.ops.calls)
        flat = torch.zeros(eng.lora_numel)
        dx = eng.backward(r_out, flat_grad=flat, accumulate=False)
        assert not eng.pool.live or all(isinstance(k, int) for k in eng.pool.live)
        res[ck] = dict(y=y, dx=dx, flat=flat, d_emb=eng.d_emb_all.clone(), pool=eng.pool.bytes, n_fwd=n_fwd,
                       n_bwd=len(eng.ops.calls) - n_fwd, sites=len(eng.drop_sites), live=len(eng.pool.live))
        # a second step on the same plan (replayed closures): same numbers again
        y2 = eng.forward_tape(x, ts, ctx, 16, tc, None, emb_all=emb_all, seed=77)
        flat2 = torch.zeros(eng.lora_numel)
        dx2 = eng.backward(r_out, flat_grad=flat2, accumulate=False)
        assert torch.equal(y2, y) and torch.equal(dx2, dx) and torch.equal(flat2, flat)
    a, b = res[False], res[True]
    assert torch.equal(a["y"], b["y"]) and torch.equal(a["dx"], b["dx"]) and torch.equal(a["flat"], b["flat"])
    assert torch.equal(a["d_emb"], b["d_emb"])
    assert float(a["flat"].abs().sum()) > 0 and a["sites"] == b["sites"] > 100
    assert b["pool"] < 0.6 * a["pool"], (a["pool"], b["pool"])
    assert a["n_fwd"] == b["n_fwd"] and b["n_bwd"] > a["n_bwd"] + 0.8 * a["n_fwd"]
    assert a["live"] == b["live"]       # nothing leaks from the discarded first forward of a block


def test_lora_branch_in_the_base_leaf_epilogue_variant_of_the_engine():
    """``fuse_lora`` (T2V_LORA_EPILOGUE=1): the up-projection and the dropout of every LoRA group ride in the base leaf's GEMM
    epilogue (t2v_gemm lora_* fields) — two launches per group instead of 2 + leaves, no M x N up-projection in memory.  Train
    mode: the masks are those of the three-launch form bit for bit, so output, d/d(latents) and every LoRA gradient agree with it
    (fp32 emulation: to rounding), and with autograd through the torch module with the masks replayed."""
    g = load("unet_tiny")
    x, ts, ctx, tc = g["x"], g["ts"], g["ctx"], g["tc"]
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(5))
    res = {}
    for fused in (False, True):
        m, params = _student("unet_tiny", 64)
        m.train()
        eng = UNetGradEngine(m, EmuOps(strict=True))
        eng.fuse_lora = fused
        eng.bind_lora(params)
        torch.manual_seed(3)
        emb_all = m.conditioning_emb_all(ts, 16, tc, None).detach()
        y = eng.forward_tape(x, ts, ctx, 16, tc, None, emb_all=emb_all, seed=77)
        n_fwd = eng.ops.calls.count("gemm")
        flat = torch.zeros(eng.lora_numel)
        dx = eng.backward(r_out, flat_grad=flat, accumulate=False)
        res[fused] = (y, dx, flat, n_fwd, len(eng.drop_sites))
    a, b = res[False], res[True]
    assert rel_l2(b[0], a[0]) < 1e-5 and rel_l2(b[1], a[1]) < 1e-4 and rel_l2(b[2], a[2]) < 1e-4
    assert a[4] == b[4] > 100
    assert b[3] < a[3] - 400, (a[3], b[3])          # one up-projection launch per leaf gone from the forward
