"""The WHOLE library — all eight translation units of libt2v_hip.so, built for the host SIMT simulator — behind the same
``native.HipOps`` wrappers and the same engines as on the GPU: no emulated op anywhere.  The inference engine's UNet forward
(record once, replay) must match the fp32 oracle fixture within the end-to-end bf16 tolerance of the GPU tests, and the
forward kernels that are hardware-validated (flash attention with its transposed scores, MFMA temporal attention, GroupNorm,
LayerNorm, the direct small-channel conv, layout conversions ...) thereby calibrate the simulator's models of
`v_mfma_f32_16x16x32_bf16`, `v_mfma_f32_16x16x16_bf16`, wave votes and the LDS-DMA in a second kernel family."""
import os
import shutil
import sys

import pytest
import torch

from oracle.synth import synth_state_dict
from t2v_turbo_amd import native as nt
from tests.emu_ops import EmuOps
from tests.util import load, manifest, rel_l2, tiny_unet_params

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim"))


@pytest.fixture(scope="module")
def full_ops():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    import build as hostsim_build
    from tests.test_hostsim_kernels import HostSimOps

    def make():
        ops = HostSimOps(hostsim_build.build_full())
        ops.tune, ops._ws = {}, {}
        return ops
    return make


def test_the_simulated_library_exports_the_whole_c_abi(full_ops):
    lib = full_ops().lib
    for name in nt.EXPORTED:
        assert hasattr(lib, name), name


def test_unet_forward_on_the_real_library_matches_the_oracle_fixture(full_ops):
    from t2v_turbo_amd.engine import UNetEngine
    from t2v_turbo_amd.unet3d import UNetModel
    g = load("unet_tiny")
    m = UNetModel(**tiny_unet_params()).eval()
    m.load_state_dict(synth_state_dict(manifest("unet_tiny")), strict=True)
    m.requires_grad_(False)
    x = g["x"][:, :, :2, :8, :8].contiguous()  # (2 frames of 8x8: ~1 min of simulation)
    with torch.no_grad():
        m.native_mode = "off"
        y_ref = m(x, g["ts"], context=g["ctx"], fps=16, timestep_cond=g["tc"])
    eng = UNetEngine(m, full_ops())
    y = eng(x, g["ts"], g["ctx"], 16, g["tc"], None)
    assert rel_l2(y, y_ref) < 3e-2
    # replay of the recorded launch list with new inputs
    x2 = torch.randn(x.shape, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y2_ref = m(x2, torch.tensor([279]), context=g["ctx"], fps=24, timestep_cond=g["tc"])
    y2 = eng(x2, torch.tensor([279]), g["ctx"], 24, g["tc"], None)
    assert len(eng.plans) == 1 and rel_l2(y2, y2_ref) < 3e-2


@pytest.mark.skipif(os.environ.get("T2V_HOSTSIM_FULL") != "1", reason="minutes of simulation: set T2V_HOSTSIM_FULL=1")
@pytest.mark.parametrize("flash", [False, True])   # True also switches the weight gradients to t2v_wgrad_tn
def test_training_step_on_the_real_library_only(full_ops, monkeypatch, flash):
    """The native student step — forward, backward, all LoRA gradients — with NOTHING emulated: every launch is real kernel
    source on the simulator, through the GPU path's record / replay protocol."""
    from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
    from tests.test_unet_lora_grad_cpu import _autograd, _student
    g = load("unet_tiny")
    m, params = _student("unet_tiny", 64)
    x, ts, ctx, tc = g["x"][:, :, :2, :8, :8].contiguous(), g["ts"], g["ctx"], g["tc"]
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(5))
    y_ref, dx_ref, g_ref = _autograd(m, params, x, ts, ctx, 16, tc, None, r_out)
    eng = UNetGradEngine(m, full_ops())
    eng.flash_attn_bwd = flash  # spatial self-attention backward: flash-style kernels (csrc/attention_bwd.hip) or the GEMM form
    eng.tn_wgrad = flash        # weight gradients: token-contracted kernel (csrc/wgrad_tn.hip) or transposes + t2v_gemm
    eng.bind_lora(params)
    for step in range(2):  # second pass: replayed lists, LoRA operand packs refreshed by the gather kernel
        emb_all = m.conditioning_emb_all(ts, 16, tc, None)
        y = eng.forward_tape(x, ts, ctx, 16, tc, None, emb_all=emb_all)
        flat = torch.zeros(eng.lora_numel)
        dx = eng.backward(r_out, flat_grad=flat, accumulate=False)
        mine = {id(p) for mod in eng.engine_leaves() for p in (mod.lora_up.weight, mod.lora_down.weight)}
        off, errs = 0, []
        for p, r in zip(params, g_ref):
            if id(p) in mine and float(r.abs().max()) > 0:
                errs.append(rel_l2(flat[off:off + p.numel()].view_as(p), r))
            off += p.numel()
        errs = torch.tensor(errs)
        print(f"step {step}: out {rel_l2(y, y_ref):.2e}, dx {rel_l2(dx, dx_ref):.2e}, LoRA gradients median {float(errs.median()):.2e} "
              f"max {float(errs.max()):.2e}")
        assert rel_l2(y, y_ref) < 3e-2 and rel_l2(dx, dx_ref) < 6e-2
        assert torch.isfinite(errs).all() and float(errs.median()) < 8e-2 and float(errs.max()) < 0.25


@pytest.mark.skipif(os.environ.get("T2V_HOSTSIM_FULL") != "1", reason="minutes of simulation: set T2V_HOSTSIM_FULL=1")
@pytest.mark.parametrize("new_kernels", [False, True])   # True: flash-style attention backward + token-contracted weight gradients
def test_train_mode_step_on_the_real_library_only(full_ops, monkeypatch, new_kernels):
    """Train-mode student (dropout kernel, per-frame text K / V) on the real library: the same seed reproduces the step bit for
    bit (the backward regenerates the forward's masks from it), another seed does not, every gradient is finite."""
    from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
    from tests.test_unet_lora_grad_cpu import _student
    g = load("unet_tiny")
    m, params = _student("unet_tiny", 64)
    m.train()
    x, ts, ctx, tc = g["x"][:, :, :2, :8, :8].contiguous(), g["ts"], g["ctx"], g["tc"]
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(5))
    eng = UNetGradEngine(m, full_ops())
    eng.flash_attn_bwd = eng.tn_wgrad = new_kernels
    eng.bind_lora(params)
    with torch.no_grad():
        emb_all = m.conditioning_emb_all(ts, 16, tc, None)
    res = []
    for seed in (11, 12, 11):
        y = eng.forward_tape(x, ts, ctx, 16, tc, None, emb_all=emb_all, seed=seed)
        flat = torch.zeros(eng.lora_numel)
        dx = eng.backward(r_out, flat_grad=flat, accumulate=False)
        assert torch.isfinite(y).all() and torch.isfinite(dx).all() and torch.isfinite(flat).all()
        res.append((y, dx, flat))
    assert torch.equal(res[0][0], res[2][0]) and torch.equal(res[0][1], res[2][1]) and torch.equal(res[0][2], res[2][2])
    assert rel_l2(res[1][0], res[0][0]) > 1e-3 and rel_l2(res[1][2], res[0][2]) > 1e-3
    assert len(eng.drop_sites) > 100


def test_vae_decode_and_encode_on_the_real_library_match_the_reference_fixtures(full_ops):
    """The VAE engines (hardware-validated) on the simulated library against the fixtures made by running the reference: a second
    model family through the simulator's GEMM / GroupNorm / softmax / small-channel conv models (stride-2 convs with the
    encoder's asymmetric padding, the single-head 512-wide attention as GEMMs)."""
    from t2v_turbo_amd.engine_vae import VAEDecodeEngine, VAEEncodeEngine
    from t2v_turbo_amd.vae import AutoencoderKL
    from tests.util import VAE_TINY_DD
    g = load("vae_tiny")
    ae = AutoencoderKL(ddconfig=VAE_TINY_DD, embed_dim=4).eval()
    ae.load_state_dict(synth_state_dict(manifest("vae_tiny")), strict=True)
    ae.requires_grad_(False)
    z = g["z"][:, :, :1].contiguous()
    v = VAEDecodeEngine(ae, full_ops()).decode_frames(z, scale=1.0 / 0.18215)
    assert v.shape == g["video"][:, :, :1].shape
    assert rel_l2(v, g["video"][:, :, :1]) < 3e-2
    e = load("vae_tiny_enc")
    img = e["x"][:1].unsqueeze(2).contiguous()                     # (b, 3, t = 1, H, W)
    mom = VAEEncodeEngine(ae, full_ops()).encode_frames(img)      # (b, 8, 1, H/8, W/8)
    ref = e["moments"][:1].unsqueeze(2)
    assert mom.shape == ref.shape and rel_l2(mom.float(), ref) < 3e-2


def test_vae_decode_gradient_engine_on_the_real_library(full_ops):
    """d(loss)/d(latents) through the VAE decoder (reward branch, hardware-validated kernels of csrc/backward.hip in engine
    context) on the simulated library vs torch autograd through the module."""
    from t2v_turbo_amd.engine_vae_bwd import VAEDecodeGradEngine
    from t2v_turbo_amd.vae import AutoencoderKL
    from tests.util import VAE_TINY_DD
    ae = AutoencoderKL(ddconfig=VAE_TINY_DD, embed_dim=4).eval()
    ae.load_state_dict(synth_state_dict(manifest("vae_tiny")), strict=True)
    ae.requires_grad_(False)
    g = torch.Generator().manual_seed(3)
    z = torch.randn(1, 4, 4, 4, generator=g)
    dout = torch.randn(1, 3, 32, 32, generator=g)
    zz = z.clone().requires_grad_(True)
    ae.native_mode = "off"
    ref = ae.decode(zz)
    (ref * dout).sum().backward()
    eng = VAEDecodeGradEngine(ae, full_ops())
    out = eng.decode_frames_tape(z.unsqueeze(2), scale=1.0).squeeze(2)
    dz = eng.backward(dout.unsqueeze(2)).squeeze(2)
    assert rel_l2(out, ref.detach()) < 3e-2
    assert rel_l2(dz, zz.grad) < 5e-2


def test_optimizer_and_scheduler_kernels_on_the_real_library(full_ops):
    """The remaining entry points of the C-ABI on the simulator: fused AdamW step (against torch.optim.AdamW), deterministic sum of
    squares, EMA, the fused LCM scheduler step and the 3-term linear combination (against the emulated definitions)."""
    from tests.emu_ops import EmuOps
    ops, emu = full_ops(), EmuOps()
    gen = torch.Generator().manual_seed(0)
    n = 5000
    p0, g = torch.randn(n, generator=gen), torch.randn(n, generator=gen)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05)
    p, m, v = p0.clone(), torch.zeros(n), torch.zeros(n)
    for step in range(1, 4):
        ref.grad = (g * step * 0.5).clone()
        opt.step()
        ops.adamw_step(p, g * step, m, v, 1e-2, 0.9, 0.99, 1e-8, 0.05, step, 0.5)   # grad_scale folds the clip coefficient in
    assert torch.allclose(p, ref.detach(), rtol=2e-5, atol=2e-6)
    ws, out = torch.zeros(1024), torch.zeros(1)
    ops.sumsq(g, ws, out)
    assert abs(float(out) - float((g.double() ** 2).sum())) < 1e-3 * float((g ** 2).sum())
    tgt, src = torch.randn(n, generator=gen), torch.randn(n, generator=gen)
    want = tgt * 0.99 + src * 0.01
    ops.ema_update(tgt, src, 0.99)
    assert torch.allclose(tgt, want, rtol=1e-6, atol=1e-7)
    x, eps_, noise = (torch.randn(1, 4, 2, 8, 8, generator=gen) for _ in range(3))
    prev_s, den_s, prev_e, den_e = (torch.zeros_like(x) for _ in range(4))
    args = (0.8, 0.6, 0.3, 0.7, 0.9, 0.43)
    ops.lcm_step(x, eps_, noise, *args, prev_s, den_s)
    emu.lcm_step(x, eps_, noise, *args, prev_e, den_e)
    assert torch.allclose(prev_s, prev_e, rtol=1e-5, atol=1e-6) and torch.allclose(den_s, den_e, rtol=1e-5, atol=1e-6)
    y, z = torch.randn_like(x), torch.randn_like(x)
    o_s, o_e = torch.zeros_like(x), torch.zeros_like(x)
    ops.lincomb3(x, y, z, [0.5], [-1.5], [2.0], o_s)
    emu.lincomb3(x, y, z, [0.5], [-1.5], [2.0], o_e)
    assert torch.allclose(o_s, o_e, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("c0,c1,units,rows,silu", [(320, 0, 2, 64, True), (128, 64, 1, 160, True), (64, 64, 3, 32, False),
                                                    (320, 0, 1, 13120, True), (64, 32, 2, 64, True), (640, 320, 2, 160, False)])
def test_group_norm_from_the_producers_column_statistics(full_ops, c0, c1, units, rows, silu):
    """t2v_group_norm_cs (csrc/norm.hip): GroupNorm whose statistics come from per-32-row column sums (what t2v_gemm's colstat_out
    writes) instead of a pass over the tensor — single tensors and virtual concats, several units, against the plain GroupNorm.
    Both forms of the statistics launch: the direct one (one block per group writes the per-channel affine: 256 threads, and 1 024
    for the one-unit 13 120-row case; a group straddling the two parts of a concat in the 128 + 64 and 640 + 320 cases) and the
    partial-sums one (64 + 32: three channels per group, odd)."""
    from tests.emu_ops import EmuOps
    sim, emu = full_ops(), EmuOps()
    gen = torch.Generator().manual_seed(c0 + rows)
    C, M = c0 + c1, units * rows
    x0 = (torch.randn(M, c0, generator=gen) * 1.3 + 0.4).bfloat16()
    x1 = (torch.randn(M, c1, generator=gen) * 0.7 - 0.2).bfloat16() if c1 else None
    gamma, beta = torch.randn(C, generator=gen) * 0.2 + 1.0, torch.randn(C, generator=gen) * 0.1

    def colstats(t):
        v = t.float().reshape(M // 32, 32, -1)
        return torch.stack([v.sum(1), (v * v).sum(1)], dim=2).contiguous()

    cs0, cs1 = colstats(x0), (colstats(x1) if c1 else None)
    out_s = torch.full((M, C), float("nan"), dtype=torch.bfloat16)
    ws = torch.zeros(max(sim.group_norm_cs_ws_floats(units, rows, 32), 1))
    sim.group_norm_cs(cs0, cs1, x0, x1, units, rows, 1e-5, gamma, beta, silu, ws, out_s)
    out_e = torch.zeros(M, C)
    emu.group_norm(x0.float(), None if x1 is None else x1.float(), units, rows, 1e-5, gamma, beta, silu, None, out_e)
    assert torch.isfinite(out_s.float()).all() and rel_l2(out_s.float(), out_e) < 4e-3
    out_e2 = torch.zeros(M, C)
    emu.group_norm_cs(cs0, cs1, x0.float(), None if x1 is None else x1.float(), units, rows, 1e-5, gamma, beta, silu, None, out_e2)
    assert rel_l2(out_e2, out_e) < 1e-5
    # t2v_gn_stats_cs: the (mean, rstd) the training engine keeps for the backward, from the same column statistics
    st_s, st_e, st_t = torch.full((units, 64), float("nan")), torch.zeros(units, 64), torch.zeros(units, 64)
    sim.gn_stats_cs(cs0, cs1, c0, c1, units, rows, 1e-5, ws, st_s)
    emu.gn_stats_cs(cs0, cs1, c0, c1, units, rows, 1e-5, None, st_e)
    emu.gn_stats(x0.float(), None if x1 is None else x1.float(), units, rows, 1e-5, None, st_t)
    assert torch.isfinite(st_s).all() and rel_l2(st_s, st_e) < 1e-5 and rel_l2(st_e, st_t) < 1e-4


@pytest.mark.parametrize("C,M", [(64, 200), (320, 250), (64, 192)])
def test_fused_feed_forward_kernel(full_ops, C, M):
    """t2v_ffn_fused (csrc/ffn.hip): LayerNorm -> GEGLU projection -> output projection -> + residual in one launch, the packed
    fragment-order weights from native.ffn_pack — against the emulation (which decodes the packed operands by the header's layout
    rules) and against the plain module arithmetic; ragged last workgroup / wave included."""
    from tests.emu_ops import EmuOps
    sim, emu = full_ops(), EmuOps()
    gen = torch.Generator().manual_seed(C + M)
    inner = 4 * C
    x = (torch.randn(M, C, generator=gen) * 1.2 + 0.3).bfloat16()
    w1 = (torch.randn(2 * inner, C, generator=gen) * C ** -0.5).bfloat16().float()
    b1 = torch.randn(2 * inner, generator=gen) * 0.1
    w2 = (torch.randn(C, inner, generator=gen) * inner ** -0.5).bfloat16().float()
    b2 = torch.randn(C, generator=gen) * 0.1
    gamma, beta = torch.randn(C, generator=gen) * 0.2 + 1.0, torch.randn(C, generator=gen) * 0.1
    assert sim.ffn_fused_supported(C)
    pk = nt.ffn_pack(w1, b1, w2, b2, gamma, beta, torch.bfloat16)
    out_s = torch.full((M, C), float("nan"), dtype=torch.bfloat16)
    sim.ffn_fused(x, *pk, 1e-5, out_s)
    out_e = torch.zeros(M, C)
    emu.ffn_fused(x.float(), *[t.float() for t in pk], 1e-5, out_e)
    xf = x.float()
    h = torch.nn.functional.layer_norm(xf, (C,), gamma, beta, 1e-5) @ w1.t() + b1
    ref = xf + (h[:, :inner] * torch.nn.functional.gelu(h[:, inner:])) @ w2.t() + b2
    assert torch.isfinite(out_s.float()).all()
    assert rel_l2(out_e, ref) < 5e-3                    # (the packed W1 diag(gamma) is rounded to bf16)
    assert rel_l2(out_s.float(), out_e) < 6e-3          # bf16 normalised rows and hidden activations inside the kernel


def test_c_side_replay_equals_the_python_loop(full_ops):
    """t2v_replay (csrc/replay.hip) walks a recorded launch list inside the library: same launches, same arguments, same order
    as the per-launch ctypes loop — bit-identical outputs — and a failing launch is reported with its index."""
    import ctypes as C
    sim = full_ops()
    x = torch.randn(37, 64).bfloat16()
    w = torch.randn(48, 64).bfloat16()
    b = torch.randn(48)
    out1, out2 = torch.empty(37, 48).bfloat16(), torch.empty(37, 48).bfloat16()
    ln1, ln2 = torch.empty(37, 48).bfloat16(), torch.empty(37, 48).bfloat16()
    g, be = torch.randn(48), torch.randn(48)
    outs = []
    for replay_c in (False, True):
        sim.recording = []
        o, l = (out2, ln2) if replay_c else (out1, ln1)
        sim.gemm(x, w, o, M=37, N=48, bias=b)
        sim.layernorm(o, g, be, 1e-5, l)
        rec, sim.recording = sim.recording, None
        o.zero_(); l.zero_()
        sim.c_replay = replay_c
        sim.replay(rec, None)
        outs.append((o.clone(), l.clone()))
        if replay_c:
            segs = sim.compile_recording(rec)
            assert [s[0] for s in segs] == ["c"] and segs[0][2] == sum(2 + len(a) for _, a, _ in rec)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert sim.lib.t2v_replay_lookup(b"t2v_gemm") >= 0 and sim.lib.t2v_replay_lookup(b"t2v_version") == -1
    # a malformed program is refused, not executed
    bad = (C.c_ulonglong * 3)(9999, 1, 0)
    failed = C.c_int(-7)
    assert sim.lib.t2v_replay(bad, 3, None, C.byref(failed)) < 0


def _rt5(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).bfloat16().float()


def _bf5(t):
    return t.to(torch.bfloat16).contiguous()


# ---------------------------------------------------------------------------------- round 5: row-group LayerNorm; GroupNorm forms at UNet widths
@pytest.mark.parametrize("M,Cc", [(70, 320), (37, 640), (19, 1280), (130, 64), (9, 128), (40, 192), (33, 1024), (5, 2048), (21, 48)])
def test_layernorm_row_group_and_fallback_widths(full_ops, M, Cc):
    """t2v_layernorm: the row-group kernels (8 / 16 / 32 / 64 lanes per row x <= 5 chunks: 320, 640, 1280, powers of two, ragged row
    counts that leave lane groups without a row) and the one-row-per-wave fallback (48 channels = 6 chunks) against the emulation."""
    sim, emu = full_ops(), EmuOps()
    x = _rt5(M, Cc, seed=M + Cc, scale=2.0) + 0.5
    g = torch.Generator().manual_seed(Cc)
    gamma, beta = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    o_s, o_e = torch.zeros(M, Cc, dtype=torch.bfloat16), torch.zeros(M, Cc)
    sim.layernorm(_bf5(x), gamma, beta, 1e-5, o_s)
    emu.layernorm(x, gamma, beta, 1e-5, o_e)
    assert rel_l2(o_s.float(), o_e) < 6e-3


@pytest.mark.parametrize("units,rows,c0,c1,silu", [(2, 96, 320, 0, True), (3, 64, 64, 128, False), (1, 2560, 320, 0, True),
                                                   (2, 40, 1280, 1280, True), (1, 72, 2560, 0, False), (1, 640, 1280, 0, True),
                                                   (3, 40, 1280, 0, False), (1, 900, 1280, 0, True), (2, 40, 1280, 256, True)])
def test_group_norm_two_and_three_launch_forms_one_and_two_chunks_per_thread(full_ops, units, rows, c0, c1, silu):
    """t2v_group_norm (statistics pass + apply whose blocks finish the statistics themselves, or the three-launch form for many slabs):
    one and two channel chunks per thread (C > 2048), virtual concats, ragged slabs — written for a variant of the apply kernel that
    issued its first rows before finishing the statistics (measured slower on MI355X and dropped: csrc/norm.hip); kept as coverage
    of the width / slab-count corners.  Round 5: where a group has a multiple of 8 channels and a unit's slice of it fits 4 chunks per
    thread, the op is ONE launch with a workgroup per (group, unit) (gn_group_kernel): 256 threads (2 x 40 x 2 560, 1 x 72 x 2 560,
    3 x 40 x 1 280), 1 024 threads (1 x 640 x 1 280), a group that straddles the two parts of a concat (1 280 + 256: 48 channels per
    group), and 1 x 900 x 1 280 as the first shape past its limit."""
    sim, emu = full_ops(), EmuOps()
    Cc = c0 + c1
    x0 = _rt5(units * rows, c0, seed=rows + c0, scale=1.5) + 0.25
    x1 = (_rt5(units * rows, c1, seed=rows + c1 + 1, scale=0.7) - 0.5) if c1 else None
    g = torch.Generator().manual_seed(Cc + rows)
    gamma, beta = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    ws = torch.zeros(max(int(sim.group_norm_ws_floats(units, rows, 32, Cc)), 1))
    o_s, o_e = torch.zeros(units * rows, Cc, dtype=torch.bfloat16), torch.zeros(units * rows, Cc)
    sim.group_norm(_bf5(x0), None if x1 is None else _bf5(x1), units, rows, 1e-5, gamma, beta, silu, ws, o_s, 32)
    emu.group_norm(x0, x1, units, rows, 1e-5, gamma, beta, silu, torch.zeros(8), o_e, 32)
    assert rel_l2(o_s.float(), o_e) < 6e-3


@pytest.mark.parametrize("n_img,h,w,cin,cout,f32", [(2, 6, 8, 16, 3, True), (1, 5, 12, 64, 4, False), (3, 3, 4, 8, 1, True), (1, 4, 16, 32, 2, False)])
def test_conv3x3_small_cout_direct(full_ops, n_img, h, w, cin, cout, f32):
    """t2v_conv3x3_small_cout (the VAE decoder's conv_out: a direct VALU conv, four pixels of a row per thread) against the emulated conv:
    image borders, rows that are one quad wide, every output-channel count, fp32 and bf16 outputs."""
    sim, emu = full_ops(), EmuOps()
    M = n_img * h * w
    x = _rt5(M, cin, seed=M + cin)
    g = torch.Generator().manual_seed(cout)
    wgt = torch.randn(cout, 9 * cin, generator=g) * (9 * cin) ** -0.5
    bias = torch.randn(cout, generator=g)
    assert sim.conv_small_cout_supported(w, cin, cout) and emu.conv_small_cout_supported(w, cin, cout)
    assert not sim.conv_small_cout_supported(w + 2, cin, cout) and not sim.conv_small_cout_supported(w, cin, 5) and not sim.conv_small_cout_supported(w, cin + 4, cout)
    o_s = torch.full((M, cout), float("nan"), dtype=torch.float32 if f32 else torch.bfloat16)
    o_e = torch.zeros(M, cout)
    sim.conv_small_cout(_bf5(x), n_img, h, w, wgt, bias, o_s)
    emu.conv_small(x, n_img, h, w, wgt, bias, o_e)
    assert torch.isfinite(o_s.float()).all()
    assert rel_l2(o_s.float(), o_e) < (1e-5 if f32 else 6e-3)


@pytest.mark.parametrize("n_img,seq_q,seq_kv,heads,kv_div", [(1, 300, 130, 1, 1), (2, 70, 77, 2, 2)])
def test_attn_spatial_forms_are_bit_identical_on_the_simulator(full_ops, n_img, seq_q, seq_kv, heads, kv_div):
    """t2v_attn_spatial_form 8 (eight waves per workgroup) and 65 (64 queries per wave, the two query sets' phases offset) against the
    product kernel on the simulator: same arithmetic per query, so bit-identical outputs — ragged query blocks, padding keys in the last
    tile, per-clip keys."""
    sim = full_ops()
    inner = heads * 64
    n_kv = n_img // kv_div
    kp = ((seq_kv + 63) // 64) * 64
    q = _bf5(_rt5(n_img * seq_q, inner, seed=11, scale=2.0))
    k = _bf5(_rt5(n_kv * seq_kv, inner, seed=12, scale=2.0))
    vt = _bf5(_rt5(n_kv * inner, kp, seed=13))
    vt[:, seq_kv:] = 1e30
    outs = []
    try:
        for form in (0, 8, 65):
            sim.lib.t2v_attn_spatial_form(form)
            o = torch.full((n_img * seq_q, inner), float("nan"), dtype=torch.bfloat16)
            sim.attn_spatial(q, k, vt, kp, o, n_img, seq_q, seq_kv, heads, kv_div, 0.125)
            outs.append(o.clone())
    finally:
        sim.lib.t2v_attn_spatial_form(0)
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
