"""-m gpu: the fused-statistics variants of t2v_gemm (csrc/gemm_fuse.hip) and t2v_group_norm_cs on MI355X, through the C-ABI,
against the torch emulation — every tile id that carries them, at toy sizes and at the UNet's own shapes:

  * row statistics of the output (the next LayerNorm's, attention.py:300-311),
  * column statistics per 32-row slab (the next GroupNorm's, openaimodel3d.py:223-254 / lvdm/basics.py:78-89),
  * LayerNorm folded into the consuming GEMM (q|k|v, the text cross-attention's q, the GEGLU projection),
  * GroupNorm finished from the column statistics (single tensors and virtual concats).

(The same checks run on the host SIMT simulator in tests/test_hostsim_gemm_fuse.py / test_hostsim_full.py.)"""
import pytest
import torch

from t2v_turbo_amd import native as nt
from tests.emu_ops import EmuOps
from tests.test_hostsim_gemm_fuse import FUSED_TILES, _ln_fold_operands, _rt
from tests.util import rel_l2

pytestmark = pytest.mark.gpu

BF16_TOL = 4e-3
EMU = EmuOps()


@pytest.fixture(scope="module")
def hip():
    from t2v_turbo_amd.native import HipOps
    h = HipOps()
    h.init()
    return h


def _d(t, dtype=torch.bfloat16):
    return None if t is None else t.cuda().to(dtype).contiguous()


@pytest.mark.parametrize("cfg", FUSED_TILES)
def test_row_statistics(hip, cfg):
    M, N, K = 2080, 320, 320
    a, w, b, res = _rt(M, K, seed=1), _rt(N, K, seed=2, scale=K ** -0.5), _rt(N, seed=3), _rt(M, N, seed=4)
    out_h = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    rs_h = torch.full((M, 2 * (N // 32) + 4), 7.0, device="cuda")
    out_e, rs_e = torch.zeros(M, N), torch.zeros(M, 2 * (N // 32) + 4)
    kw = dict(M=M, N=N)
    assert hip.gemm_fuse_supported(_d(a), _d(w), out_h, bias=_d(b, torch.float32), residual=_d(res), rowstat=rs_h, tile_cfg=cfg, **kw)
    hip.gemm(_d(a), _d(w), out_h, bias=_d(b, torch.float32), residual=_d(res), rowstat=rs_h, tile_cfg=cfg, **kw)
    EMU.gemm(a, w, out_e, bias=b, residual=res, rowstat=rs_e, **kw)
    torch.cuda.synchronize()
    nb = N // 32
    assert rel_l2(out_h.float().cpu(), out_e) < BF16_TOL
    assert rel_l2(rs_h[:, :2 * nb].cpu(), rs_e[:, :2 * nb]) < 1e-4
    assert float(rs_h[:, 2 * nb:].min()) == 7.0 and float(rs_h[:, 2 * nb:].max()) == 7.0


@pytest.mark.parametrize("cfg", FUSED_TILES)
def test_column_statistics(hip, cfg):
    n, h, w, c0, N = 4, 16, 16, 128, 320
    M = n * h * w
    x, wt, b, rv = _rt(M, c0, seed=1), _rt(N, 9 * c0, seed=2, scale=(9 * c0) ** -0.5), _rt(N, seed=3), _rt(n, N, seed=4)
    out_h = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    cs_h = torch.full((M // 32, N, 2), float("nan"), device="cuda")
    out_e, cs_e = torch.zeros(M, N), torch.zeros(M // 32, N, 2)
    kw = dict(M=M, N=N, mode=nt.GEMM_CONV3X3, n_img=n, h=h, wd=w, rowvec_div=h * w)
    hk = dict(kw, tile_cfg=cfg, split_k=1)   # (a 24-tile launch with K = 1152: the heuristic would split K, and a split launch carries no statistics)
    assert hip.gemm_fuse_supported(_d(x), _d(wt), out_h, bias=_d(b, torch.float32), rowvec=_d(rv, torch.float32), colstat=cs_h, **hk)
    hip.gemm(_d(x), _d(wt), out_h, bias=_d(b, torch.float32), rowvec=_d(rv, torch.float32), colstat=cs_h, **hk)
    EMU.gemm(x, wt, out_e, bias=b, rowvec=rv, colstat=cs_e, **kw)
    torch.cuda.synchronize()
    assert rel_l2(out_h.float().cpu(), out_e) < BF16_TOL
    stored = out_h.float().cpu().reshape(M // 32, 32, N)
    want = torch.stack([stored.sum(1), (stored * stored).sum(1)], dim=2)
    got = cs_h.cpu()
    assert torch.isfinite(got).all() and rel_l2(got, want) < 1e-5
    assert rel_l2(got, cs_e) < 2e-2


# (GEGLU needs 64-wide wave tiles in N: tile ids 5, 9, 23, 31 do not carry it)
@pytest.mark.parametrize("cfg,C,N,geglu", [(cfg, C, N, g) for cfg in FUSED_TILES
                                           for C, N, g in ((320, 960, False), (640, 512, True), (1280, 256, False), (512, 1536, False))
                                           if not (g and cfg in (5, 9, 23, 31))])
def test_layernorm_fold(hip, cfg, C, N, geglu):
    M = 1000
    x, wp, s_vec, t_vec, ref = _ln_fold_operands(M, C, N, seed=10 + C, geglu=geglu)
    nb = C // 32
    xb = x.reshape(M, nb, 32)
    stats = torch.zeros(M, 2 * nb + 4)
    stats[:, :2 * nb] = torch.stack([xb.sum(2), (xb * xb).sum(2)], dim=2).reshape(M, -1)
    act = nt.ACT_GEGLU if geglu else nt.ACT_NONE
    n_out = N // 2 if geglu else N
    out_h = torch.full((M, n_out), float("nan"), dtype=torch.bfloat16, device="cuda")
    out_e = torch.zeros(M, n_out)
    kw = dict(M=M, N=N, act=act)
    lnf_h = (stats.cuda(), 1e-5, s_vec.cuda())
    hk = dict(kw, tile_cfg=cfg, split_k=1)   # (few tiles and a deep K: the heuristic would split K, and a split launch carries no fold)
    assert hip.gemm_fuse_supported(_d(x), _d(wp), out_h, bias=t_vec.cuda(), lnf=lnf_h, **hk)
    hip.gemm(_d(x), _d(wp), out_h, bias=t_vec.cuda(), lnf=lnf_h, **hk)
    EMU.gemm(x, wp, out_e, bias=t_vec, lnf=(stats, 1e-5, s_vec), **kw)
    torch.cuda.synchronize()
    got = out_h.float().cpu()
    assert torch.isfinite(got).all() and rel_l2(got, out_e) < BF16_TOL
    assert rel_l2(out_e, ref) < 6e-3


def test_producer_consumer_chain_at_the_unet_shapes(hip):
    """The real chain at the 320-channel level (M = 40960): out-projection with residual + row statistics -> q|k|v with the
    LayerNorm folded in, on the tiles the tuned table picks, against LayerNorm -> Linear in fp32."""
    M, C = 40960, 320
    gen = torch.Generator().manual_seed(0)
    o = torch.randn(M, C, generator=gen).bfloat16().float()
    y = (torch.randn(M, C, generator=gen) * 2.0 + 0.5).bfloat16().float()
    Wo, bo = _rt(C, C, seed=1, scale=C ** -0.5), _rt(C, seed=2)
    gamma, beta = _rt(C, seed=3) * 0.2 + 1.0, _rt(C, seed=4) * 0.1
    Wqkv = _rt(3 * C, C, seed=5, scale=C ** -0.5)
    y1 = torch.zeros(M, C, dtype=torch.bfloat16, device="cuda")
    rs = torch.zeros(M, C // 16, device="cuda")
    hip.gemm(_d(o), _d(Wo), y1, M=M, N=C, bias=_d(bo, torch.float32), residual=_d(y), rowstat=rs)
    wp = (Wqkv * gamma[None, :]).bfloat16()
    s_vec, t_vec = wp.float().sum(1), Wqkv @ beta
    qkv = torch.zeros(M, 3 * C, dtype=torch.bfloat16, device="cuda")
    assert hip.gemm_fuse_supported(y1, wp.cuda(), qkv, M=M, N=3 * C, bias=t_vec.cuda(), lnf=(rs, 1e-5, s_vec.cuda()))
    hip.gemm(y1, wp.cuda(), qkv, M=M, N=3 * C, bias=t_vec.cuda(), lnf=(rs, 1e-5, s_vec.cuda()))
    torch.cuda.synchronize()
    y1_ref = o @ Wo.t() + bo + y
    ref = torch.nn.functional.layer_norm(y1_ref, (C,), gamma, beta, 1e-5) @ Wqkv.t()
    assert rel_l2(y1.float().cpu(), y1_ref) < BF16_TOL
    assert rel_l2(qkv.float().cpu(), ref) < 8e-3      # two bf16 roundings (y1, W diag(gamma)) ahead of the product


@pytest.mark.parametrize("c0,c1,units,rows,silu", [(320, 0, 16, 2560, True), (320, 0, 1, 40960, True), (1280, 640, 16, 160, True),
                                                   (640, 0, 1, 10240, False), (1280, 1280, 1, 640, True)])
def test_group_norm_from_column_statistics(hip, c0, c1, units, rows, silu):
    gen = torch.Generator().manual_seed(c0 + rows)
    C, M = c0 + c1, units * rows
    x0 = (torch.randn(M, c0, generator=gen) * 1.3 + 0.4).bfloat16()
    x1 = (torch.randn(M, c1, generator=gen) * 0.7 - 0.2).bfloat16() if c1 else None
    gamma, beta = torch.randn(C, generator=gen) * 0.2 + 1.0, torch.randn(C, generator=gen) * 0.1

    def colstats(t):
        v = t.float().reshape(M // 32, 32, -1)
        return torch.stack([v.sum(1), (v * v).sum(1)], dim=2).contiguous()

    cs0, cs1 = colstats(x0), (colstats(x1) if c1 else None)
    out_h = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device="cuda")
    ws = torch.zeros(max(hip.group_norm_cs_ws_floats(units, rows, 32), 1), device="cuda")
    hip.group_norm_cs(cs0.cuda(), None if cs1 is None else cs1.cuda(), x0.cuda(), None if x1 is None else x1.cuda(), units, rows, 1e-5,
                      gamma.cuda(), beta.cuda(), silu, ws, out_h)
    out_e = torch.zeros(M, C)
    EMU.group_norm(x0.float(), None if x1 is None else x1.float(), units, rows, 1e-5, gamma, beta, silu, None, out_e)
    torch.cuda.synchronize()
    got = out_h.float().cpu()
    assert torch.isfinite(got).all() and rel_l2(got, out_e) < BF16_TOL
    # deterministic: a second call gives the same bits
    out2 = torch.zeros_like(out_h)
    hip.group_norm_cs(cs0.cuda(), None if cs1 is None else cs1.cuda(), x0.cuda(), None if x1 is None else x1.cuda(), units, rows, 1e-5,
                      gamma.cuda(), beta.cuda(), silu, ws, out2)
    torch.cuda.synchronize()
    assert torch.equal(out2, out_h)
    # t2v_gn_stats_cs: (mean, rstd) per (unit, group) for the training engine's backward, from the same column statistics
    st_h = torch.full((units, 64), float("nan"), device="cuda")
    st_t = torch.zeros(units, 64, device="cuda")
    hip.gn_stats_cs(cs0.cuda(), None if cs1 is None else cs1.cuda(), c0, c1, units, rows, 1e-5, ws, st_h)
    ws2 = torch.zeros(max(hip.gn_ws_floats(units, rows, 32), 1), device="cuda")
    hip.gn_stats(x0.cuda(), None if x1 is None else x1.cuda(), units, rows, 1e-5, ws2, st_t)
    torch.cuda.synchronize()
    assert torch.isfinite(st_h).all() and rel_l2(st_h.cpu(), st_t.cpu()) < 1e-4


@pytest.mark.parametrize("C,M", [(64, 200), (320, 250), (320, 40960)])
def test_fused_feed_forward(hip, C, M):
    """t2v_ffn_fused (csrc/ffn.hip) on MI355X: out = x + FF(LayerNorm(x)) in one launch, against the emulation (decoding the packed
    operands) and the plain arithmetic of attention.py:300-311,516-542."""
    from t2v_turbo_amd import native as _nt
    if not _nt.has_experimental():
        pytest.skip("entry point of a T2V_EXPERIMENTAL=1 build (python t2v-turbo_amd/csrc/build.py with T2V_EXPERIMENTAL=1, then T2V_HIP_LIB=.../libt2v_hip_exp.so)")

    gen = torch.Generator().manual_seed(C + M)
    inner = 4 * C
    x = (torch.randn(M, C, generator=gen) * 1.2 + 0.3).bfloat16()
    w1 = (torch.randn(2 * inner, C, generator=gen) * C ** -0.5).bfloat16().float()
    b1 = torch.randn(2 * inner, generator=gen) * 0.1
    w2 = (torch.randn(C, inner, generator=gen) * inner ** -0.5).bfloat16().float()
    b2 = torch.randn(C, generator=gen) * 0.1
    gamma, beta = torch.randn(C, generator=gen) * 0.2 + 1.0, torch.randn(C, generator=gen) * 0.1
    assert hip.ffn_fused_supported(C)
    pk = nt.ffn_pack(w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda(), gamma.cuda(), beta.cuda(), torch.bfloat16)
    out_h = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device="cuda")
    hip.ffn_fused(x.cuda(), *pk, 1e-5, out_h)
    torch.cuda.synchronize()
    xf = x.float()
    h = torch.nn.functional.layer_norm(xf, (C,), gamma, beta, 1e-5) @ w1.t() + b1
    ref = xf + (h[:, :inner] * torch.nn.functional.gelu(h[:, inner:])) @ w2.t() + b2
    got = out_h.float().cpu()
    assert torch.isfinite(got).all() and rel_l2(got, ref) < 8e-3
    out2 = torch.zeros_like(out_h)
    hip.ffn_fused(x.cuda(), *pk, 1e-5, out2)
    torch.cuda.synchronize()
    assert torch.equal(out2, out_h)


def _dv(t):
    return None if t is None else t.bfloat16().cuda().contiguous()


@pytest.mark.parametrize("M,K,C,leaves", [(40960, 320, 320, 3), (40960, 320, 320, 1), (10240, 640, 640, 1), (2560, 1280, 1280, 3), (1000, 128, 96, 3)])
def test_lora_branch_in_the_base_leaf_epilogue(hip, M, K, C, leaves):
    """t2v_gemm lora_* fields (FUSE bit 16): y = x W^T + b + residual + s * dropout(t U^T) in one launch, one- and three-leaf groups at
    the UNet's shapes, with and without the dropout mask, against the emulation; deterministic.  First run on MI355X in the last
    seconds of round 3's GPU budget (tools/r3_gpu_calls/r3_call22.sh): green."""
    N, p, site = leaves * C, 0.1, 4
    seed = torch.tensor([0x5EED_1234_ABCD], dtype=torch.int64)
    x, w, b = _rt(M, K, seed=1), _rt(N, K, seed=2, scale=K ** -0.5), _rt(N, seed=3)
    res, t, u = _rt(M, N, seed=4), _rt(M, leaves * 64, seed=5, scale=0.5), _rt(N, 64, seed=6, scale=0.2)
    for drop in (None, (p, seed, site, N, 0)):
        o_h, o_e = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda"), torch.zeros(M, N)
        kw = dict(M=M, N=N, bias=b)
        drop_h = None if drop is None else (p, seed.cuda(), site, N, 0)
        lo_h = (_dv(t), _dv(u), C, 0.5)
        assert hip.gemm_fuse_supported(_dv(x), _dv(w), o_h, residual=_dv(res), lora=lo_h, bias=b.cuda(), M=M, N=N, dropout=drop_h)
        hip.gemm(_dv(x), _dv(w), o_h, residual=_dv(res), lora=lo_h, bias=b.cuda(), M=M, N=N, dropout=drop_h)
        EMU.gemm(x, w, o_e, residual=res, lora=(t, u, C, 0.5), dropout=drop, **kw)
        torch.cuda.synchronize()
        got = o_h.float().cpu()
        assert torch.isfinite(got).all() and rel_l2(got, o_e) < BF16_TOL, (drop is not None, rel_l2(got, o_e))
        o2 = torch.zeros_like(o_h)
        hip.gemm(_dv(x), _dv(w), o2, residual=_dv(res), lora=lo_h, bias=b.cuda(), M=M, N=N, dropout=drop_h)
        torch.cuda.synchronize()
        assert torch.equal(o2, o_h)
