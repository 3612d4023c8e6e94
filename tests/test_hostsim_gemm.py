"""t2v_gemm — the MFMA implicit-GEMM kernel source itself — on the host SIMT simulator (tests/hostsim): `v_mfma_f32_32x32x16_bf16`
and the LDS-DMA are modelled as wave collectives (operand / accumulator lane layouts of the CDNA4 ISA), `s_waitcnt lgkmcnt(0)`
and compiler fences as wave rendezvous points (a wave runs in lockstep on the hardware), counted vmcnt waits as no-ops.

1. Calibration: the 23 tile configurations that ARE validated on MI355X reproduce the emulated backend on the shapes of the GPU
   tests (linear / virtual concat / every conv gather mode / masking / short K) — so the model of the matrix instruction, of
   the DMA's lane-linear LDS image, of the XOR swizzle and of the slab epilogue is right.
2. Then what has NOT run on hardware: tile ids 24-29 (4-wave 256x256 with 128x128 wave tiles; the register-staged operand path),
   and the operand / epilogue combinations only the LoRA training path produces (K = tokens with deep forced split-K, 1- and
   4-row problems, N = 4 bf16 output with a residual of row stride 8, column-sliced operands and outputs, batched per-head
   launches with zero batch strides, alpha with fp32 output).

The simulator says nothing about speed, register pressure or the memory model — only that the arithmetic and addressing of
the source are right."""
import os
import shutil
import sys

import pytest
import torch

from t2v_turbo_amd import native as nt
from tests.emu_ops import EmuOps
from tests.util import rel_l2

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim"))

BF16_TOL = 4e-3
VALIDATED = list(range(1, 24))
EXPERIMENTAL = list(range(24, 30))


def _rt(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).bfloat16().float()


@pytest.fixture(scope="module")
def sim():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    import build as hostsim_build
    from tests.test_hostsim_kernels import HostSimOps
    ops = HostSimOps(hostsim_build.build_gemm())
    ops.tune, ops._ws = {}, {}
    return ops


EMU = EmuOps()


def _case(sim, *, M, N, c0, c1=0, mode=0, n_img=0, h=0, w=0, frames=0, bias=True, rowvec_div=0, residual=False, act=0, alpha=1.0,
          out_f32=False, cfg=0, rows=None, seed=0, split=0):
    taps = {nt.GEMM_LINEAR: 1, nt.GEMM_TCONV3: 3}.get(mode, 9)
    K = taps * (c0 + c1)
    rows = rows or M
    a0 = _rt(rows, c0, seed=seed)
    a1 = _rt(rows, c1, seed=seed + 1) if c1 else None
    wt = _rt(N, K, seed=seed + 2, scale=K ** -0.5)
    b = _rt(N, seed=seed + 3) if bias else None
    n_out = N // 2 if act == nt.ACT_GEGLU else N
    rv = _rt((M + rowvec_div - 1) // rowvec_div, n_out, seed=seed + 4) if rowvec_div else None
    res = _rt(M, n_out, seed=seed + 5) if residual else None
    out_s = torch.full((M, n_out), float("nan"), dtype=torch.float32 if out_f32 else torch.bfloat16)
    out_e = torch.zeros(M, n_out)
    kw = dict(M=M, N=N, mode=mode, n_img=n_img, h=h, wd=w, frames=frames, rowvec_div=rowvec_div, act=act, alpha=alpha)
    bf = lambda t: None if t is None else t.bfloat16().contiguous()  # noqa: E731
    sim.gemm(bf(a0), bf(wt), out_s, a1=bf(a1), bias=b, rowvec=rv, residual=bf(res), tile_cfg=cfg, split_k=split, **kw)
    EMU.gemm(a0, wt, out_e, a1=a1, bias=b, rowvec=rv, residual=res, **kw)
    got = out_s.float()
    assert torch.isfinite(got).all(), "kernel left output elements unwritten / non-finite"
    return rel_l2(got, out_e)


def _tile_suite(sim, cfg):
    n, h, w = 2, 6, 7
    conv = dict(n_img=n, h=h, w=w, rows=n * h * w, cfg=cfg)
    errs = [
        _case(sim, M=300, N=320, c0=320, residual=True, cfg=cfg),
        _case(sim, M=200, N=192, c0=128, c1=64, cfg=cfg, seed=3),                       # virtual concat
        _case(sim, M=77, N=64, c0=256, bias=False, cfg=cfg, seed=5),
        _case(sim, M=200, N=128, c0=64, cfg=cfg, seed=7),                               # K shorter than the DMA ring
        _case(sim, M=130, N=96, c0=64, c1=128, cfg=cfg, seed=9),
        _case(sim, M=n * h * w, N=128, c0=64, mode=nt.GEMM_CONV3X3, rowvec_div=h * w, residual=True, **conv),
        _case(sim, M=n * 3 * 4, N=64, c0=128, mode=nt.GEMM_CONV3X3_S2, seed=2, **conv),
        _case(sim, M=n * 4 * h * w, N=64, c0=64, mode=nt.GEMM_CONV3X3_UP2, seed=3, **conv),
        _case(sim, M=n * 3 * 4, N=64, c0=64, mode=nt.GEMM_CONV3X3_S2_PAD01, seed=4, n_img=n, h=6, w=8, rows=n * 48, cfg=cfg),
        _case(sim, M=2 * 4 * 15, N=128, c0=128, mode=nt.GEMM_TCONV3, n_img=8, h=3, w=5, frames=4, rows=120, residual=True, cfg=cfg, seed=6),
    ]
    return max(errs)


@pytest.mark.parametrize("cfg", VALIDATED)
def test_calibration_on_hardware_validated_tiles(sim, cfg):
    assert _tile_suite(sim, cfg) < BF16_TOL


@pytest.mark.parametrize("cfg", EXPERIMENTAL)
def test_experimental_tiles_not_yet_run_on_hardware(sim, cfg):
    assert _tile_suite(sim, cfg) < BF16_TOL
    if cfg != 24:  # GEGLU needs 64-wide wave tiles in N (id 24 has 128)
        assert _case(sim, M=150, N=256, c0=128, act=nt.ACT_GEGLU, cfg=cfg, seed=11) < BF16_TOL


def test_epilogues_and_split_k(sim):
    assert _case(sim, M=300, N=512, c0=128, act=nt.ACT_GEGLU) < BF16_TOL
    assert _case(sim, M=130, N=256, c0=64, act=nt.ACT_SILU, seed=2) < BF16_TOL
    assert _case(sim, M=200, N=4, c0=64, out_f32=True, seed=4) < 2e-3
    assert _case(sim, M=200, N=3, c0=128, seed=6) < BF16_TOL
    assert _case(sim, M=256, N=128, c0=64, alpha=0.125, bias=False, seed=8) < BF16_TOL
    assert _case(sim, M=2, N=1280, c0=320, act=nt.ACT_SILU, seed=9) < BF16_TOL
    for split in (2, 3, 5):
        assert _case(sim, M=2 * 5 * 8, N=256, c0=256, mode=nt.GEMM_CONV3X3, n_img=2, h=5, w=8, residual=True, rowvec_div=40, split=split) < BF16_TOL
        assert _case(sim, M=100, N=64, c0=1280, act=nt.ACT_SILU, out_f32=True, split=split, seed=3) < 2e-3


# ---------------------------------------------------------------------------------- what only the LoRA training path asks of t2v_gemm
@pytest.mark.parametrize("M,N,K,split", [(320, 64, 2560, 40), (4, 64, 640, 4), (576, 320, 1280, 16), (1, 320, 640, 4), (192, 64, 128, 1)])
def test_weight_gradient_shapes(sim, M, N, K, split):
    """dU = s dy^T t, dD = G^T x, the per-clip column sums: a few output tiles, K = tokens, deep forced split-K, fp32 out, alpha."""
    a, w = _rt(M, K, seed=1, scale=0.2), _rt(N, K, seed=2, scale=0.2)
    o_e, o_s = torch.zeros(M, N), torch.full((M, N), float("nan"))
    EMU.gemm(a, w, o_e, M=M, N=N, alpha=0.5)
    sim.gemm(a.bfloat16(), w.bfloat16(), o_s, M=M, N=N, alpha=0.5, split_k=split)
    assert rel_l2(o_s, o_e) < 2e-3


def test_lora_branch_operand_views(sim):
    """z = s t U^T + residual into a column slice of a padded buffer (N = 4: the exit conv's LoRA branch, row stride 8); g = s dy U
    from a column slice of dy into a column slice of g; the weight-gradient GEMM reading row blocks of the transposed operands and
    writing a column slice of the fp32 arena."""
    M = 150
    t, U, R = _rt(M, 192, seed=1), _rt(4, 64, seed=2), _rt(M, 8, seed=3)
    for dev in ("emu", "sim"):
        tt, uu, rr = (t, U, R) if dev == "emu" else (t.bfloat16(), U.bfloat16(), R.bfloat16())
        zf = torch.zeros(M, 8, dtype=tt.dtype)
        (EMU if dev == "emu" else sim).gemm(tt[:, 64:128], uu, zf[:, :4], M=M, N=4, alpha=0.5, residual=rr[:, :4])
        if dev == "emu":
            z_e = zf.clone()
    assert rel_l2(zf.float()[:, :4], z_e[:, :4]) < BF16_TOL and float(zf.float()[:, 4:].abs().max()) == 0
    dy, UT = _rt(M, 384, seed=4), _rt(64, 128, seed=5)
    g_e, g_s = torch.zeros(M, 192), torch.zeros(M, 192, dtype=torch.bfloat16)
    EMU.gemm(dy[:, 128:256], UT, g_e[:, 64:128], M=M, N=64, alpha=2.0)
    sim.gemm(dy.bfloat16()[:, 128:256], UT.bfloat16(), g_s[:, 64:128], M=M, N=64, alpha=2.0)
    assert rel_l2(g_s.float(), g_e) < BF16_TOL
    dyT, tT = _rt(384, 192, seed=6), _rt(192, 192, seed=7)   # [N_total, Mp], [n*rp, Mp]
    E_e, E_s = torch.zeros(128, 256), torch.zeros(128, 256)
    EMU.gemm(dyT[128:256], tT[64:128], E_e[:, 64:128], M=128, N=64, alpha=0.5)
    sim.gemm(dyT.bfloat16()[128:256], tT.bfloat16()[64:128], E_s[:, 64:128], M=128, N=64, alpha=0.5, split_k=3)
    assert rel_l2(E_s, E_e) < 2e-3


def test_per_head_batches_of_the_attention_backward(sim):
    """Batched launches with the engine's stride patterns: heads as the inner batch index with a ZERO outer stride (the frames of a
    clip share K / V), column offsets as batch strides on A / W / out."""
    heads, mq, L, kp = 3, 96, 77, 128
    q, k = _rt(mq, heads * 64, seed=1, scale=0.3), _rt(L, heads * 64, seed=2, scale=0.3)
    s_e, s_s = torch.zeros(heads * mq, kp), torch.zeros(heads * mq, kp, dtype=torch.bfloat16)
    kw = dict(M=mq, N=L, alpha=0.125, batch=heads, batch_inner=heads, a_strides=(0, 64), w_strides=(0, 64), o_strides=(0, mq * kp))
    EMU.gemm(q[:, :64], k[:, :64], s_e, **kw)
    sim.gemm(q.bfloat16()[:, :64], k.bfloat16()[:, :64], s_s, **kw)
    assert rel_l2(s_s.float(), s_e) < BF16_TOL and float(s_s.float()[:, L:].abs().max()) == 0
    # dV[kv][c] = sum_q P[q][kv] dO[q][c]: transposed operands per head, token-major output columns per head, M = 77 rows
    pT, doT = _rt(heads * kp, 128, seed=3, scale=0.2), _rt(heads * 64, 128, seed=4)
    dv_e, dv_s = torch.zeros(L, heads * 64), torch.zeros(L, heads * 64, dtype=torch.bfloat16)
    kw = dict(M=L, N=64, batch=heads, batch_inner=heads, a_strides=(0, kp * 128), w_strides=(0, 64 * 128), o_strides=(0, 64))
    EMU.gemm(pT, doT, dv_e[:, :64], **kw)
    sim.gemm(pT.bfloat16(), doT.bfloat16(), dv_s[:, :64], split_k=2, **kw)
    assert rel_l2(dv_s.float(), dv_e) < BF16_TOL


@pytest.mark.skipif(os.environ.get("T2V_HOSTSIM_FULL") != "1", reason="minutes of simulation: set T2V_HOSTSIM_FULL=1")
def test_training_step_with_the_real_gemm_kernel_on_the_simulator():
    """The native student step with EVERY t2v_gemm launch (implicit-GEMM convs, LoRA branch, token-contracted weight gradients,
    split-K, batched attention-backward products) executed by the real kernel source on the simulator, next to the simulated SIMT
    kernels; only forward attention / norms / layout ops are emulated.  bf16 bounds of the opt-in device test."""
    from t2v_turbo_amd.engine_unet_bwd import UNetGradEngine
    from tests.hybrid_ops import HybridOps
    from tests.test_unet_lora_grad_cpu import _autograd, _student
    from tests.util import load
    g = load("unet_tiny")
    m, params = _student("unet_tiny", 64)
    x, ts, ctx, tc = g["x"][:, :, :2, :8, :8].contiguous(), g["ts"], g["ctx"], g["tc"]
    r_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(5))
    y_ref, dx_ref, g_ref = _autograd(m, params, x, ts, ctx, 16, tc, None, r_out)
    ops_h = HybridOps(real_gemm=True)
    eng = UNetGradEngine(m, ops_h)
    eng.bind_lora(params)
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
    print(f"real-gemm launches {ops_h.gemm_calls}, simulated SIMT launches {ops_h.sim_calls}: out {rel_l2(y, y_ref):.2e}, "
          f"dx {rel_l2(dx, dx_ref):.2e}, LoRA gradients median {float(errs.median()):.2e} max {float(errs.max()):.2e}")
    assert rel_l2(y, y_ref) < 3e-2 and rel_l2(dx, dx_ref) < 6e-2
    assert torch.isfinite(errs).all() and float(errs.median()) < 8e-2 and float(errs.max()) < 0.25


@pytest.mark.parametrize("cfg", [1, 4, 9, 18])
def test_dropout_epilogue_matches_the_standalone_mask(sim, cfg):
    """t2v_gemm's dropout fields (the LoRA up-projection's epilogue, utils/lora.py:45-50): the mask must be t2v_dropout_bf16's on
    the launch's column block — zeros exactly where the emulated mask drops, kept values scaled by 1 / (1 - p), residual added
    behind the mask; two leaves of one group write two column blocks of the same masked matrix."""
    M, K, p, site = 200, 64, 0.25, 5
    seed = torch.tensor([0x1234_5678_9ABC], dtype=torch.int64)
    a = _rt(M, K, seed=1)
    res = (_rt(M, 320 + 128, seed=9).abs() + 1.0).bfloat16().float()   # strictly positive: dropped positions show as res exactly
    z_s = torch.zeros(M, 320 + 128, dtype=torch.bfloat16)
    z_e = torch.zeros(M, 320 + 128)
    c0 = 0
    for N, sd in ((320, 2), (128, 3)):
        wt = _rt(N, K, seed=sd, scale=K ** -0.5)
        drop = (p, seed, site, 320 + 128, c0)
        sim.gemm(a.bfloat16(), wt.bfloat16().contiguous(), z_s[:, c0:c0 + N], M=M, N=N, alpha=0.5, residual=res[:, c0:c0 + N].bfloat16(),
                 tile_cfg=cfg, dropout=drop)
        EMU.gemm(a, wt, z_e[:, c0:c0 + N], M=M, N=N, alpha=0.5, residual=res[:, c0:c0 + N], dropout=drop)
        c0 += N
    keep = EMU.dropout_keep(int(seed[0]), site, M, 320 + 128, p)
    got = z_s.float()
    assert rel_l2(got, z_e) < BF16_TOL
    assert torch.equal(got[~keep], res[~keep])            # dropped: the residual alone, bit for bit
    assert abs(float(keep.float().mean()) - (1 - p)) < 0.01


@pytest.mark.parametrize("M,K,res,rowvec", [(320, 320, True, 0), (200, 64, False, 0), (480, 1280, True, 160)])
def test_layernorm_second_output_of_the_160x320_tile(sim, M, K, res, rowvec):
    """t2v_gemm's ln_* fields: LayerNorm(out) over the 320 columns as a second output of the full-row 160x320 kernel (two waves
    per row exchange their row sums through LDS) — main output unchanged, LN output = F.layer_norm of the fp32 epilogue values;
    rows beyond M masked; inputs with |mean| >> std (a residual stream is not zero-mean)."""
    N = 320
    a = _rt(M, K, seed=1)
    wt = _rt(N, K, seed=2, scale=K ** -0.5)
    b = _rt(N, seed=3)
    r = (_rt(M, N, seed=5) * 0.5 + 6.0).bfloat16().float() if res else None
    rv = _rt((M + rowvec - 1) // rowvec, N, seed=4) if rowvec else None
    gamma, beta = _rt(N, seed=6) * 0.1 + 1.0, _rt(N, seed=7) * 0.1
    bf = lambda t: None if t is None else t.bfloat16().contiguous()  # noqa: E731
    out_s, ln_s = torch.full((M, N), float("nan"), dtype=torch.bfloat16), torch.full((M, N), float("nan"), dtype=torch.bfloat16)
    out_p = torch.zeros(M, N, dtype=torch.bfloat16)
    out_e, ln_e = torch.zeros(M, N), torch.zeros(M, N)
    kw = dict(M=M, N=N, rowvec_div=rowvec)
    sim.gemm(bf(a), bf(wt), out_s, bias=b, rowvec=rv, residual=bf(r), ln=(gamma, beta, 1e-5, ln_s), **kw)
    sim.gemm(bf(a), bf(wt), out_p, bias=b, rowvec=rv, residual=bf(r), tile_cfg=23, split_k=1, **kw)
    EMU.gemm(a, wt, out_e, bias=b, rowvec=rv, residual=r, ln=(gamma, beta, 1e-5, ln_e), **kw)
    assert torch.equal(out_s, out_p)                      # the main output is the plain kernel's, bit for bit
    assert torch.isfinite(ln_s.float()).all()
    assert rel_l2(out_s.float(), out_e) < BF16_TOL and rel_l2(ln_s.float(), ln_e) < BF16_TOL
