"""FULL fine-tuning of the student UNet on the native gradient engine's dataflow (CPU, emulated op backend, fp32): the gradient of EVERY
parameter, d(loss)/d(latents) and d(loss)/d(emb_all) against torch autograd through the reference-shaped module — the call pattern of
train_latent_t2v_turbo_v2.py (:669 ``unet.requires_grad_(True)``, :798-816 every parameter in an optimizer group, :1262 backward).  Pins:
the token-contracted weight gradients of Linear leaves (q | k | v groups split per leaf, virtual-concat inputs split per part, GEGLU's
row permutation undone), the im2col matrices of every conv gather mode (3x3, stride 2, nearest-x2, (3,1,1), the 4-channel entry conv
and the 4-channel exit conv) and their tap-major -> parameter-layout gather, bias / GroupNorm(+SiLU) / LayerNorm affine gradients,
per-layer text K / V projections, the per-clip column sums that carry d(loss)/d(emb_all), and the in-place pack refresh after an
optimizer step (Packer.refresh) under an unchanged launch plan."""
import warnings

import torch

from oracle.synth import synth_state_dict
from t2v_turbo_amd.unet3d import UNetModel
from tests.emu_ops import EmuOps
from tests.util import load, manifest, rel_l2, tiny_unet_params


def _student(fixture="unet_tiny", **cfg):
    m = UNetModel(**tiny_unet_params(**cfg))
    m.load_state_dict(synth_state_dict(manifest(fixture)), strict=True)
    gen = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for p in m.parameters():   # zero-initialised output projections would make most gradients vanish
            if float(p.abs().max()) == 0:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.05)
    m.requires_grad_(True)
    m.eval()   # (dropout masks are the engine's counter-based ones in train mode: compared separately on the device with replayed masks)
    return m


def _grads(m, route, x, ts, ctx, tc, r_out):
    for p in m.parameters():
        p.grad = None
    xg = x.clone().requires_grad_(True)
    m.native_mode = route
    y = m(xg, ts, context=ctx, fps=16, timestep_cond=tc)
    (y * r_out).sum().backward()
    return y.detach(), xg.grad.clone(), {n: (None if p.grad is None else p.grad.clone()) for n, p in m.named_parameters()}


def _compare(got, ref, tol=3e-4):
    worst = (0.0, None)
    for n, r in ref.items():
        g = got[n]
        assert (g is None) == (r is None), n
        if r is None:
            continue
        if float(r.abs().max()) == 0:
            assert float(g.abs().max()) < 1e-6, n
            continue
        e = rel_l2(g, r)
        if e > worst[0]:
            worst = (e, n)
    assert worst[0] < tol, worst


def test_every_parameter_gradient_matches_autograd_and_survives_an_optimizer_step():
    g = load("unet_tiny")
    m = _student()
    m._native_ops_factory = EmuOps
    x, ts, ctx, tc = g["x"], g["ts"], g["ctx"], g["tc"]
    r_out = torch.randn(g["y"].shape, generator=torch.Generator().manual_seed(3))
    y_ref, dx_ref, ref = _grads(m, "off", x, ts, ctx, tc, r_out)
    with warnings.catch_warnings():
        warnings.simplefilter("error")   # the route must not fall back to the torch composite (it warns when it does)
        assert m._auto_route(x.clone().requires_grad_(True), ctx, tc, None)[0] == "train_full"
        y, dx, got = _grads(m, "train", x, ts, ctx, tc, r_out)
    eng = m._engine_box.full
    assert eng is not None and eng.training_full and len(eng.plans) == 1
    assert rel_l2(y, y_ref) < 2e-5 and rel_l2(dx, dx_ref) < 3e-4
    assert all(v is not None for v in ref.values())
    _compare(got, ref)
    # an optimizer step moves every weight: the SAME plan must give the gradients of the new weights (packs re-filled in place)
    plan = next(iter(eng.plans.values()))
    with torch.no_grad():
        gen = torch.Generator().manual_seed(5)
        for p in m.parameters():
            p.add_(torch.randn(p.shape, generator=gen) * 0.02 * float(p.abs().mean() + 1e-3))
    y_ref2, dx_ref2, ref2 = _grads(m, "off", x, ts, ctx, tc, r_out)
    y2, dx2, got2 = _grads(m, "train", x, ts, ctx, tc, r_out)
    assert next(iter(eng.plans.values())) is plan and len(eng.plans) == 1
    assert rel_l2(y_ref2, y_ref) > 1e-3, "the weight update must change the output for this check to mean anything"
    assert rel_l2(y2, y_ref2) < 2e-5 and rel_l2(dx2, dx_ref2) < 3e-4
    _compare(got2, ref2)


def test_partially_frozen_network_and_auto_route():
    """requires_grad on a subset (the v2 script's temporal / other parameter groups can be trained separately): frozen parameters get no
    gradient, the others are unchanged; the auto route takes full fine-tuning only when a parameter is trainable."""
    g = load("unet_tiny")
    m = _student()
    m._native_ops_factory = EmuOps
    x, ts, ctx, tc = g["x"], g["ts"], g["ctx"], g["tc"]
    r_out = torch.randn(g["y"].shape, generator=torch.Generator().manual_seed(4))
    for n, p in m.named_parameters():
        p.requires_grad_("temporal" in n or "temopral" in n or n.startswith("time_embed"))
    _, dx_ref, ref = _grads(m, "off", x, ts, ctx, tc, r_out)
    _, dx, got = _grads(m, "train", x, ts, ctx, tc, r_out)
    assert any(v is None for v in ref.values()) and any(v is not None for v in ref.values())
    assert rel_l2(dx, dx_ref) < 3e-4
    _compare(got, ref)
    m.requires_grad_(False)
    assert m._auto_route(x.clone().requires_grad_(True), ctx, tc, None)[0] == "composite"   # input gradients only: not this route


def _fixture_step(m, x, ts, ctx, tc, r_out, route):
    """(output, d/d latents, gradients in named_parameters order) through ``route``."""
    y, dx, grads = _grads(m, route, x, ts, ctx, tc, r_out)
    return y, dx, [grads[n] for n, _ in m.named_parameters()]


def check_against_reference_fixture(y, dx, grads, names, gg, out_tol, dx_tol, norm_tol, proj_tol, full_tol):
    """Compare one step with tests/golden/unet_tiny_full_grad.npz / unet_mid_full_grad.npz (made by the imported reference:
    make_golden_full_grad.py; the mid-width fixture holds digests only)."""
    from tests.golden.make_golden_full_grad import KEEP_FULL, digests
    if "g_" + KEEP_FULL[0].replace(".", "__") not in gg:
        KEEP_FULL = ()
    assert [str(n) for n in gg["names"]] == names, "parameter registration order differs from the reference's"
    e_out, e_dx = rel_l2(y, gg["out"]), rel_l2(dx, gg["dx"])
    d, ref = torch.from_numpy(digests(grads)), gg["digests"]
    norm_err = (d[:, 0] - ref[:, 0]).abs() / ref[:, 0]
    proj_err = ((d[:, 1:] - ref[:, 1:]).abs() / ref[:, :1]).max(dim=1).values
    print(f"[full fine-tuning fixture] out {e_out:.3e} dx {e_dx:.3e}; per-parameter norm err max {float(norm_err.max()):.4f} median "
          f"{float(norm_err.median()):.5f}; projection err / norm max {float(proj_err.max()):.4f} median {float(proj_err.median()):.5f}", flush=True)
    assert e_out < out_tol and e_dx < dx_tol
    assert float(norm_err.max()) < norm_tol, names[int(norm_err.argmax())]
    assert float(proj_err.max()) < proj_tol[0] and float(proj_err.median()) < proj_tol[1], names[int(proj_err.argmax())]
    for n in KEEP_FULL:
        assert rel_l2(grads[names.index(n)], gg["g_" + n.replace(".", "__")]) < full_tol, n


import pytest


@pytest.mark.parametrize("fixture,width", [("unet_tiny_full_grad", 64), ("unet_mid_full_grad", 128)])
def test_module_autograd_reproduces_the_reference_full_gradient_fixture(fixture, width):
    """The checker of the engine tests — autograd through this repository's torch module — against the REFERENCE's own parameter
    gradients (tests/golden/unet_tiny_full_grad.npz, and unet_mid_full_grad.npz: the reference at model_channels = 128): same
    registration order, every gradient to fp32 round-off."""
    from oracle.synth import manifest_of
    from tests.golden.make_golden_full_grad import SEED_R
    g, gg = load("unet_tiny"), load(fixture)
    m = UNetModel(**tiny_unet_params(model_channels=width))
    m.load_state_dict(synth_state_dict(manifest_of(m)), strict=True)
    m.requires_grad_(True)
    m.eval()
    r_out = torch.randn(g["x"].shape, generator=torch.Generator().manual_seed(SEED_R))
    assert torch.equal(r_out, gg["r_out"])
    y, dx, grads = _fixture_step(m, g["x"], g["ts"], g["ctx"], g["tc"], r_out, "off")
    check_against_reference_fixture(y, dx, grads, [n for n, _ in m.named_parameters()], gg, 1e-5, 1e-4, 1e-4, (1e-3, 1e-4), 1e-4)
