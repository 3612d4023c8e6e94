"""The fused-statistics variants of t2v_gemm (csrc/gemm_fuse.hip) on the host SIMT simulator: row statistics for the next
LayerNorm, column statistics per 32-row slab for the next GroupNorm, and the LayerNorm folded into the consuming GEMM
(attention.py:300-311 / openaimodel3d.py:223-254 without the normalisation passes) — every tile id that carries them, against
the emulated backend, before any of it ran on hardware.  The calibration of the simulator is tests/test_hostsim_gemm.py."""
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
FUSED_TILES = [4, 5, 7, 9, 11, 12, 16, 17, 19, 20, 23, 31]
EMU = EmuOps()


def _rt(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).bfloat16().float()


def _bf(t):
    return None if t is None else t.bfloat16().contiguous()


@pytest.fixture(scope="module")
def sim():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    import build as hostsim_build
    from tests.test_hostsim_kernels import HostSimOps
    ops = HostSimOps(hostsim_build.build_gemm())
    ops.tune, ops._ws = {}, {}
    return ops


@pytest.mark.parametrize("cfg", FUSED_TILES)
def test_row_statistics_of_the_output(sim, cfg):
    """Producer of a LayerNorm input: out-projection with bias and residual (fast kernel: the accumulators start there)."""
    M, N, K = 200, 320, 128
    a, w, b, res = _rt(M, K, seed=1), _rt(N, K, seed=2, scale=K ** -0.5), _rt(N, seed=3), _rt(M, N, seed=4)
    out_s = torch.full((M, N), float("nan"), dtype=torch.bfloat16)
    rs_s = torch.full((M, 2 * (N // 32) + 4), 7.0)
    out_e, rs_e = torch.zeros(M, N), torch.zeros(M, 2 * (N // 32) + 4)
    kw = dict(M=M, N=N, bias=b)
    assert sim.gemm_fuse_supported(_bf(a), _bf(w), out_s, residual=_bf(res), rowstat=rs_s, tile_cfg=cfg, **kw)
    sim.gemm(_bf(a), _bf(w), out_s, residual=_bf(res), rowstat=rs_s, tile_cfg=cfg, **kw)
    EMU.gemm(a, w, out_e, residual=res, rowstat=rs_e, **kw)
    assert rel_l2(out_s.float(), out_e) < BF16_TOL
    nb = N // 32
    assert rel_l2(rs_s[:, :2 * nb], rs_e[:, :2 * nb]) < 1e-4
    assert float(rs_s[:, 2 * nb:].min()) == 7.0 and float(rs_s[:, 2 * nb:].max()) == 7.0   # nothing written past the blocks
    # the statistics give the LayerNorm of the stored rows
    st = rs_s[:, :2 * nb].reshape(M, nb, 2)
    mean = st[:, :, 0].sum(1) / N
    var = st[:, :, 1].sum(1) / N - mean * mean
    assert torch.allclose(mean, out_e.mean(1), atol=2e-3) and torch.allclose(var, out_e.var(1, unbiased=False), rtol=2e-2, atol=1e-3)


@pytest.mark.parametrize("cfg", FUSED_TILES)
def test_column_statistics_per_slab(sim, cfg):
    """Producer of a GroupNorm input: 3x3 conv with the time-embedding row vector, and a (3,1,1) conv with a residual; the
    statistics must be those of the bf16 values the kernel stored (what the consumer reads), slab by slab."""
    n, h, w, c0, N = 2, 8, 8, 64, 192          # M = 128: four slabs
    M = n * h * w
    x, wt, b = _rt(M, c0, seed=1), _rt(N, 9 * c0, seed=2, scale=(9 * c0) ** -0.5), _rt(N, seed=3)
    rv = _rt(n, N, seed=4)
    out_s = torch.full((M, N), float("nan"), dtype=torch.bfloat16)
    cs_s = torch.full((M // 32, N, 2), float("nan"))
    out_e, cs_e = torch.zeros(M, N), torch.zeros(M // 32, N, 2)
    kw = dict(M=M, N=N, mode=nt.GEMM_CONV3X3, n_img=n, h=h, wd=w, bias=b, rowvec_div=h * w)
    assert sim.gemm_fuse_supported(_bf(x), _bf(wt), out_s, rowvec=rv, colstat=cs_s, tile_cfg=cfg, **kw)
    sim.gemm(_bf(x), _bf(wt), out_s, rowvec=rv, colstat=cs_s, tile_cfg=cfg, **kw)
    EMU.gemm(x, wt, out_e, rowvec=rv, colstat=cs_e, **kw)
    assert rel_l2(out_s.float(), out_e) < BF16_TOL
    stored = out_s.float().reshape(M // 32, 32, N)
    want = torch.stack([stored.sum(1), (stored * stored).sum(1)], dim=2)
    assert torch.isfinite(cs_s).all() and rel_l2(cs_s, want) < 1e-5      # exactly the stored values' sums (fp32 order aside)
    assert rel_l2(cs_s, cs_e) < 2e-2
    # temporal conv + residual, M = 2 clips x 4 frames x 8 pixels = 64 rows, N not a multiple of the wave tile
    F, hw, C = 4, 8, 64
    M2 = 2 * F * hw
    x2, w2, r2 = _rt(M2, C, seed=5), _rt(320, 3 * C, seed=6, scale=(3 * C) ** -0.5), _rt(M2, 320, seed=7)
    o2 = torch.full((M2, 320), float("nan"), dtype=torch.bfloat16)
    cs2 = torch.full((M2 // 32, 320, 2), float("nan"))
    kw2 = dict(M=M2, N=320, mode=nt.GEMM_TCONV3, n_img=2 * F, h=2, wd=4, frames=F)
    sim.gemm(_bf(x2), _bf(w2), o2, residual=_bf(r2), colstat=cs2, tile_cfg=cfg, **kw2)
    st2 = o2.float().reshape(M2 // 32, 32, 320)
    assert rel_l2(cs2, torch.stack([st2.sum(1), (st2 * st2).sum(1)], dim=2)) < 1e-5


def _ln_fold_operands(M, C, N, seed, geglu=False):
    """Raw rows x, LayerNorm affine, consumer Linear -> what the folded launch takes and what the unfolded reference gives."""
    x = _rt(M, C, seed=seed) * 1.5 + 0.3
    gamma, beta = _rt(C, seed=seed + 1) * 0.2 + 1.0, _rt(C, seed=seed + 2) * 0.1
    W, b = _rt(N, C, seed=seed + 3, scale=C ** -0.5), _rt(N, seed=seed + 4)
    wp = (W * gamma[None, :]).bfloat16().float()           # W diag(gamma), as packed (bf16)
    s_vec = wp.sum(dim=1)                                   # row sums of the PACKED weights
    t_vec = b + W @ beta
    ref = torch.nn.functional.layer_norm(x, (C,), gamma, beta, 1e-5) @ W.t() + b
    if geglu:   # packed [32 value | 32 gate] column groups: emulate on the packed columns directly
        g = ref.reshape(M, N // 64, 2, 32)
        ref = (g[:, :, 0] * torch.nn.functional.gelu(g[:, :, 1])).reshape(M, N // 2)
    return x, wp, s_vec, t_vec, ref


# (GEGLU needs 64-wide wave tiles in N: tile ids 5, 9, 23, 31 do not carry it)
@pytest.mark.parametrize("cfg,C,N,geglu", [(cfg, C, N, g) for cfg in FUSED_TILES for C, N, g in ((320, 192, False), (128, 256, True), (512, 128, False))
                                           if not (g and cfg in (5, 9, 23, 31))])
def test_layernorm_folded_into_the_consumer(sim, cfg, C, N, geglu):
    M = 200
    x, wp, s_vec, t_vec, ref = _ln_fold_operands(M, C, N, seed=10 + C, geglu=geglu)
    # the producer's statistics: any launch that writes x with rowstat — here an identity GEMM would do; take the emulation
    nb = C // 32
    xb = x.reshape(M, nb, 32)
    stats = torch.zeros(M, 2 * nb + 4)
    stats[:, :2 * nb] = torch.stack([xb.sum(2), (xb * xb).sum(2)], dim=2).reshape(M, -1)
    act = nt.ACT_GEGLU if geglu else nt.ACT_NONE
    n_out = N // 2 if geglu else N
    out_s = torch.full((M, n_out), float("nan"), dtype=torch.bfloat16)
    out_e = torch.zeros(M, n_out)
    kw = dict(M=M, N=N, bias=t_vec, act=act, lnf=(stats, 1e-5, s_vec))
    assert sim.gemm_fuse_supported(_bf(x), _bf(wp), out_s, tile_cfg=cfg, **kw)
    sim.gemm(_bf(x), _bf(wp), out_s, tile_cfg=cfg, **kw)
    EMU.gemm(x, wp, out_e, **kw)
    assert torch.isfinite(out_s.float()).all()
    assert rel_l2(out_s.float(), out_e) < BF16_TOL
    assert rel_l2(out_e, ref) < 6e-3          # the folded form IS LayerNorm -> Linear (up to the bf16 rounding of W diag(gamma))


@pytest.mark.parametrize("cfg", FUSED_TILES)
def test_dropout_epilogue_on_the_fast_kernels(sim, cfg):
    """The LoRA up-projection with alpha = 1 (utils/lora.py:45-50, scale 1): its dropout epilogue rides on the fast kernels' staged
    epilogue (FUSE bit 8: mask on the product, then the residual tile) — zeros exactly where the emulated mask of
    t2v_dropout_bf16 drops, kept values scaled by 1 / (1 - p), residual behind the mask; two leaves of one group write two column
    blocks of the same masked matrix; ragged M."""
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
        sim.gemm(_bf(a), _bf(wt), z_s[:, c0:c0 + N], M=M, N=N, residual=_bf(res[:, c0:c0 + N]), tile_cfg=cfg, split_k=1, dropout=drop)
        EMU.gemm(a, wt, z_e[:, c0:c0 + N], M=M, N=N, residual=res[:, c0:c0 + N], dropout=drop)
        c0 += N
    keep = EMU.dropout_keep(int(seed[0]), site, M, 320 + 128, p)
    got = z_s.float()
    assert rel_l2(got, z_e) < BF16_TOL
    assert torch.equal(got[~keep], res[~keep])            # dropped: the residual alone, bit for bit
    # without a residual, with a bias inside the mask
    b = _rt(320, seed=4)
    o_s, o_e = torch.full((M, 320), float("nan"), dtype=torch.bfloat16), torch.zeros(M, 320)
    wt = _rt(320, K, seed=2, scale=K ** -0.5)
    sim.gemm(_bf(a), _bf(wt), o_s, M=M, N=320, bias=b, tile_cfg=cfg, split_k=1, dropout=(p, seed, 7, 320, 0))
    EMU.gemm(a, wt, o_e, M=M, N=320, bias=b, dropout=(p, seed, 7, 320, 0))
    k2 = EMU.dropout_keep(int(seed[0]), 7, M, 320, p)
    assert rel_l2(o_s.float(), o_e) < BF16_TOL and float(o_s.float()[~k2].abs().max()) == 0.0


@pytest.mark.parametrize("cfg", FUSED_TILES)
def test_lora_branch_in_the_base_leaf_epilogue(sim, cfg):
    """t2v_gemm lora_* fields (csrc/gemm_fuse.hip, FUSE bit 16): y = x W^T + b + residual + s * dropout(t U^T) in ONE launch after the
    rank-64 down-projection — LoraInjectedLinear.forward (utils/lora.py:45-50) without the M x N up-projection ever in memory.  A
    three-leaf group (q | k | v: block-diagonal U, t of leaf l at columns [64 l, 64 l + 64)), the mask that of t2v_dropout_bf16
    over the group's [M][3 N] matrix (bit-identical to the three-launch form), with and without dropout, with column statistics,
    a 3x3 conv leaf with the time-embedding row vector; ragged M."""
    M, K, C, p, site = 200, 128, 96, 0.25, 4
    seed = torch.tensor([0x5EED_1234_ABCD], dtype=torch.int64)
    N = 3 * C
    x, w, b = _rt(M, K, seed=1), _rt(N, K, seed=2, scale=K ** -0.5), _rt(N, seed=3)
    res = _rt(M, N, seed=4)
    t = _rt(M, 3 * 64, seed=5, scale=0.5)
    u = _rt(N, 64, seed=6, scale=0.2)
    for drop in (None, (p, seed, site, N, 0)):
        o_s, o_e = torch.full((M, N), float("nan"), dtype=torch.bfloat16), torch.zeros(M, N)
        kw = dict(M=M, N=N, bias=b, dropout=drop)
        lo_s, lo_e = (_bf(t), _bf(u), C, 0.5), (t, u, C, 0.5)
        assert sim.gemm_fuse_supported(_bf(x), _bf(w), o_s, residual=_bf(res), lora=lo_s, tile_cfg=cfg, split_k=1, **kw)
        sim.gemm(_bf(x), _bf(w), o_s, residual=_bf(res), lora=lo_s, tile_cfg=cfg, split_k=1, **kw)
        EMU.gemm(x, w, o_e, residual=res, lora=lo_e, **kw)
        assert torch.isfinite(o_s.float()).all() and rel_l2(o_s.float(), o_e) < BF16_TOL
        # = the three-launch form of the engine: z = residual + dropout(t_l U_l^T) * s per leaf, then the base leaf with z as residual
        z = torch.zeros(M, N)
        for l in range(3):
            EMU.gemm(t[:, 64 * l:64 * l + 64], u[l * C:(l + 1) * C], z[:, l * C:(l + 1) * C], M=M, N=C, alpha=0.5,
                     residual=res[:, l * C:(l + 1) * C], dropout=None if drop is None else (p, seed, site, N, l * C))
        o3 = torch.zeros(M, N)
        EMU.gemm(x, w, o3, M=M, N=N, bias=b, residual=z)
        assert rel_l2(o_e, o3) < 1e-5
    # single leaf + column statistics for the GroupNorm that follows (the ResBlock convs): 3x3 conv, row vector, M = 128
    n, h, wd, c0, Nc = 2, 8, 8, 64, 192
    Mc = n * h * wd
    xc, wc, bc, rv = _rt(Mc, c0, seed=11), _rt(Nc, 9 * c0, seed=12, scale=(9 * c0) ** -0.5), _rt(Nc, seed=13), _rt(n, Nc, seed=14)
    tc, uc = _rt(Mc, 64, seed=15, scale=0.5), _rt(Nc, 64, seed=16, scale=0.2)
    o_s, o_e = torch.full((Mc, Nc), float("nan"), dtype=torch.bfloat16), torch.zeros(Mc, Nc)
    cs_s, cs_e = torch.full((Mc // 32, Nc, 2), float("nan")), torch.zeros(Mc // 32, Nc, 2)
    kw = dict(M=Mc, N=Nc, mode=nt.GEMM_CONV3X3, n_img=n, h=h, wd=wd, bias=bc, rowvec_div=h * wd, dropout=(p, seed, 9, Nc, 0))
    assert sim.gemm_fuse_supported(_bf(xc), _bf(wc), o_s, rowvec=rv, colstat=cs_s, lora=(_bf(tc), _bf(uc), Nc, 1.0), tile_cfg=cfg, **kw)
    sim.gemm(_bf(xc), _bf(wc), o_s, rowvec=rv, colstat=cs_s, lora=(_bf(tc), _bf(uc), Nc, 1.0), tile_cfg=cfg, **kw)
    EMU.gemm(xc, wc, o_e, rowvec=rv, colstat=cs_e, lora=(tc, uc, Nc, 1.0), **kw)
    assert rel_l2(o_s.float(), o_e) < BF16_TOL
    st = o_s.float().reshape(Mc // 32, 32, Nc)
    assert rel_l2(cs_s, torch.stack([st.sum(1), (st * st).sum(1)], dim=2)) < 1e-5
    keep = EMU.dropout_keep(int(seed[0]), 9, Mc, Nc, p)
    o_plain = torch.zeros(Mc, Nc)
    EMU.gemm(xc, wc, o_plain, rowvec=rv, **{k: v for k, v in kw.items() if k != "dropout"})
    assert rel_l2(o_s.float()[~keep], o_plain[~keep]) < BF16_TOL      # dropped positions: the base leaf alone


def test_lora_epilogue_refuses_a_wave_tile_over_three_leaves(sim):
    """The fused epilogue holds the rank-64 rows of at most two leaves per wave tile: three 32-wide leaves under the 160-wide wave
    tile of id 23 are refused (the engine then runs the three-launch form), the same group on a 32-wide wave tile is taken."""
    M, K, C = 64, 128, 32
    N = 3 * C
    x, w = _bf(_rt(M, K, seed=1)), _bf(_rt(N, K, seed=2))
    t, u = _bf(_rt(M, 3 * 64, seed=3)), _bf(_rt(N, 64, seed=4))
    out = torch.zeros(M, N, dtype=torch.bfloat16)
    assert not sim.gemm_fuse_supported(x, w, out, M=M, N=N, lora=(t, u, C, 1.0), tile_cfg=23, split_k=1)
    narrow = [c for c in FUSED_TILES if c != 23 and sim.gemm_fuse_supported(x, w, out, M=M, N=N, lora=(t, u, C, 1.0), tile_cfg=c, split_k=1)]
    assert narrow, "no fused tile takes 32-wide leaves"


def test_unsupported_requests_are_refused_not_ignored(sim):
    M, N, K = 64, 64, 64
    a, w = _bf(_rt(M, K)), _bf(_rt(N, K))
    out32 = torch.zeros(M, N)                                   # fp32 output: generic epilogue, no fused statistics
    cs = torch.zeros(M // 32, N, 2)
    assert not sim.gemm_fuse_supported(a, w, out32, M=M, N=N, colstat=cs)
    with pytest.raises(nt.NativeError):
        sim.gemm(a, w, out32, M=M, N=N, colstat=cs)
    out = torch.zeros(M, N, dtype=torch.bfloat16)
    a4, w4 = _bf(_rt(M, 4 * K)), _bf(_rt(N, 4 * K))
    assert sim.gemm_fuse_supported(a4, w4, out, M=M, N=N, colstat=cs)
    assert not sim.gemm_fuse_supported(a4, w4, out, M=M, N=N, colstat=cs, split_k=2)   # split-K: the reduce kernel finishes, not the tile


def test_gemm_plan_reports_what_the_launch_would_do(sim):
    """t2v_gemm_plan: tile id and K splits resolved as the launch resolves them, nothing launched.  A long-K conv on few rows splits K
    (given a workspace), the same launch with a dropout / LoRA epilogue is pinned to one split — the fact the gradient engine's choice of
    the LoRA form rests on (engine._lora_epilogue_pays)."""
    n, h, wd, c0, N = 2, 8, 8, 640, 128
    M = n * h * wd
    x, w = _bf(_rt(M, c0, seed=1)), _bf(_rt(N, 9 * c0, seed=2, scale=0.01))
    out = torch.zeros(M, N, dtype=torch.bfloat16)
    kw = dict(M=M, N=N, mode=nt.GEMM_CONV3X3, n_img=n, h=h, wd=wd)
    cfg, splits = sim.gemm_plan(x, w, out, **kw)
    assert cfg > 0 and splits > 1, (cfg, splits)
    cfg1, one = sim.gemm_plan(x, w, out, split_k=1, **kw)
    assert one == 1
    t, u = _bf(_rt(M, 64, seed=3)), _bf(_rt(N, 64, seed=4))
    seed = torch.tensor([7], dtype=torch.int64)
    _, s_lora = sim.gemm_plan(x, w, out, lora=(t, u, N, 1.0), dropout=(0.1, seed, 3, N, 0), **kw)
    assert s_lora == 1
    before = out.clone()
    assert torch.equal(out, before)   # (nothing ran)
