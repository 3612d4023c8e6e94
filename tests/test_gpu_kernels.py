"""-m gpu: every HIP kernel, called through the C-ABI, against the torch emulation of the same op
(tests/emu_ops.py) on identical bf16-rounded inputs.  Tolerances: outputs are bf16 (8 mantissa
bits) of fp32-accumulated values -> rel-L2 <= 4e-3 per op (fp32 outputs: <= 2e-3, limited by the
bf16 inputs of MFMA products); statistics (fp32): 1e-4."""
import os

import pytest
import torch

from tests.emu_ops import EmuOps
from tests.util import rel_l2

pytestmark = pytest.mark.gpu

BF16_TOL = 4e-3

# tile ids every shape may be tuned to, all validated on hardware; T2V_TEST_EXPERIMENTAL_TILES=1 adds the ids that are so
# far only compile-verified (24: 4-wave 256x256 with 128x128 wave tiles, 25-29: register-staged operand path)
CFGS = list(range(1, 24)) + [30, 31, 32, 33] + (list(range(24, 30)) if os.environ.get("T2V_TEST_EXPERIMENTAL_TILES") == "1" else [])


def _rt(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).bfloat16().float()


class Pair:
    """Runs one op on both backends: tensors are created from the same CPU fp32 (bf16-exact) data."""

    def __init__(self):
        from t2v_turbo_amd.native import HipOps
        self.hip, self.emu = HipOps(), EmuOps()
        self.hip.init()

    def act(self, t):  # activation / weight operand
        return t.cuda().bfloat16().contiguous(), t.clone().float().contiguous()

    def f32(self, t):
        return t.cuda().float().contiguous(), t.clone().float().contiguous()

    def run(self, name, args_hip, args_emu, kw_hip=None, kw_emu=None):
        getattr(self.hip, name)(*args_hip, **(kw_hip or {}))
        getattr(self.emu, name)(*args_emu, **(kw_emu or {}))
        torch.cuda.synchronize()


@pytest.fixture(scope="module")
def pair():
    return Pair()


def _gemm_case(pair, *, M, N, c0, c1=0, mode=0, n_img=0, h=0, w=0, frames=0, bias=True, rowvec_div=0, residual=False,
               act=0, alpha=1.0, out_f32=False, cfg=0, rows=None, seed=0, split=0):
    from t2v_turbo_amd import native as nt
    taps = {nt.GEMM_LINEAR: 1, nt.GEMM_TCONV3: 3}.get(mode, 9)
    K = taps * (c0 + c1)
    rows = rows or M
    a0 = pair.act(_rt(rows, c0, seed=seed))
    a1 = pair.act(_rt(rows, c1, seed=seed + 1)) if c1 else (None, None)
    wt = pair.act(_rt(N, K, seed=seed + 2, scale=K ** -0.5))
    b = pair.f32(_rt(N, seed=seed + 3)) if bias else (None, None)
    n_out = N // 2 if act == nt.ACT_GEGLU else N
    rv = pair.f32(_rt((M + rowvec_div - 1) // rowvec_div, n_out, seed=seed + 4)) if rowvec_div else (None, None)
    res = pair.act(_rt(M, n_out, seed=seed + 5)) if residual else (None, None)
    odt = torch.float32 if out_f32 else torch.bfloat16
    out_h = torch.full((M, n_out), float("nan"), dtype=odt, device="cuda")
    out_e = torch.zeros(M, n_out)
    kw = dict(M=M, N=N, mode=mode, n_img=n_img, h=h, wd=w, frames=frames, rowvec_div=rowvec_div, act=act, alpha=alpha)
    pair.hip.lib.t2v_gemm_force_config(cfg)
    pair.hip.lib.t2v_gemm_force_split(split)
    try:
        pair.hip.gemm(a0[0], wt[0], out_h, a1=a1[0], bias=b[0], rowvec=rv[0], residual=res[0], **kw)
    finally:
        pair.hip.lib.t2v_gemm_force_config(0)
        pair.hip.lib.t2v_gemm_force_split(0)
    pair.emu.gemm(a0[1], wt[1], out_e, a1=a1[1], bias=b[1], rowvec=rv[1], residual=res[1], **kw)
    torch.cuda.synchronize()
    got = out_h.float().cpu()
    assert torch.isfinite(got).all(), "kernel left output elements unwritten / non-finite"
    return rel_l2(got, out_e)


@pytest.mark.parametrize("cfg", CFGS)
def test_gemm_linear_tiles_and_masking(pair, cfg):
    assert _gemm_case(pair, M=300, N=320, c0=320, residual=True, cfg=cfg) < BF16_TOL
    assert _gemm_case(pair, M=1024, N=192, c0=128, c1=64, cfg=cfg, seed=3) < BF16_TOL  # virtual concat
    assert _gemm_case(pair, M=77, N=64, c0=1024, bias=False, cfg=cfg, seed=5) < BF16_TOL
    # K shorter than the DMA ring (1, 2 and 3 steps of 64): prologue / drain paths of the pipelined loop
    assert _gemm_case(pair, M=200, N=128, c0=64, cfg=cfg, seed=7) < BF16_TOL
    assert _gemm_case(pair, M=200, N=128, c0=128, cfg=cfg, seed=8) < BF16_TOL
    assert _gemm_case(pair, M=130, N=96, c0=64, c1=128, cfg=cfg, seed=9) < BF16_TOL


def test_gemm_epilogues(pair):
    from t2v_turbo_amd import native as nt
    assert _gemm_case(pair, M=500, N=512, c0=128, act=nt.ACT_GEGLU) < BF16_TOL
    assert _gemm_case(pair, M=130, N=256, c0=64, act=nt.ACT_SILU, seed=2) < BF16_TOL
    assert _gemm_case(pair, M=200, N=4, c0=64, out_f32=True, seed=4) < 2e-3          # tiny N, scalar stores
    assert _gemm_case(pair, M=200, N=3, c0=128, seed=6) < BF16_TOL
    assert _gemm_case(pair, M=256, N=128, c0=64, alpha=0.125, bias=False, seed=8) < BF16_TOL
    assert _gemm_case(pair, M=2, N=1280, c0=320, act=nt.ACT_SILU, seed=9) < BF16_TOL  # M = batch rows


@pytest.mark.parametrize("cfg", CFGS)
def test_gemm_conv_modes(pair, cfg):
    from t2v_turbo_amd import native as nt
    n, h, w = 3, 10, 12
    common = dict(n_img=n, h=h, w=w, rows=n * h * w, cfg=cfg)
    assert _gemm_case(pair, M=n * h * w, N=128, c0=64, mode=nt.GEMM_CONV3X3, rowvec_div=h * w, residual=True, **common) < BF16_TOL
    assert _gemm_case(pair, M=n * h * w, N=64, c0=64, c1=128, mode=nt.GEMM_CONV3X3, seed=1, **common) < BF16_TOL
    assert _gemm_case(pair, M=n * 5 * 6, N=64, c0=128, mode=nt.GEMM_CONV3X3_S2, seed=2, **common) < BF16_TOL
    assert _gemm_case(pair, M=n * 4 * h * w, N=64, c0=64, mode=nt.GEMM_CONV3X3_UP2, seed=3, **common) < BF16_TOL
    assert _gemm_case(pair, M=n * 5 * 6, N=64, c0=64, mode=nt.GEMM_CONV3X3_S2_PAD01, seed=4, **common) < BF16_TOL
    # odd sizes: 5x7 grid, stride 2 -> 3x4
    assert _gemm_case(pair, M=2 * 3 * 4, N=64, c0=64, mode=nt.GEMM_CONV3X3_S2, n_img=2, h=5, w=7, rows=70, cfg=cfg, seed=5) < BF16_TOL
    # temporal (3,1,1): 2 clips x 4 frames x (3x5) pixels
    assert _gemm_case(pair, M=2 * 4 * 15, N=128, c0=128, mode=nt.GEMM_TCONV3, n_img=8, h=3, w=5, frames=4, rows=120,
                      residual=True, cfg=cfg, seed=6) < BF16_TOL


def test_gemm_long_k_and_full_size_shapes(pair):
    from t2v_turbo_amd import native as nt
    # K = 9*1280 = 11520 (the 5x8 / 10x16 levels), M not a tile multiple
    assert _gemm_case(pair, M=16 * 5 * 8, N=1280, c0=1280, mode=nt.GEMM_CONV3X3, n_img=16, h=5, w=8, seed=1) < BF16_TOL
    # top level 320->320 conv on 16x40x64 tokens
    assert _gemm_case(pair, M=16 * 40 * 64, N=320, c0=320, mode=nt.GEMM_CONV3X3, n_img=16, h=40, w=64,
                      rowvec_div=16 * 40 * 64, seed=2) < BF16_TOL
    assert _gemm_case(pair, M=40960, N=2560, c0=320, act=nt.ACT_GEGLU, seed=3) < BF16_TOL


@pytest.mark.parametrize("split", [2, 3, 5])
def test_gemm_split_k(pair, split):
    from t2v_turbo_amd import native as nt
    # deep-K conv on few tokens (the 5x8 level), concat + temporal + batch-free linear, each split over K
    assert _gemm_case(pair, M=2 * 5 * 8, N=256, c0=256, mode=nt.GEMM_CONV3X3, n_img=2, h=5, w=8, residual=True,
                      rowvec_div=40, split=split) < BF16_TOL
    assert _gemm_case(pair, M=2 * 5 * 8, N=128, c0=128, c1=192, mode=nt.GEMM_CONV3X3, n_img=2, h=5, w=8, split=split, seed=1) < BF16_TOL
    assert _gemm_case(pair, M=2 * 4 * 6, N=128, c0=320, mode=nt.GEMM_TCONV3, n_img=8, h=2, w=3, frames=4, rows=48,
                      split=split, seed=2) < BF16_TOL
    assert _gemm_case(pair, M=100, N=64, c0=1280, act=nt.ACT_SILU, out_f32=True, split=split, seed=3) < 2e-3


def test_gemm_geglu_all_tiles(pair):
    from t2v_turbo_amd import native as nt
    for cfg in CFGS:
        assert _gemm_case(pair, M=300, N=256, c0=128, act=nt.ACT_GEGLU, cfg=cfg, seed=cfg) < BF16_TOL


def test_gemm_batched_two_level_strides(pair):
    # out[z0,z1] = A[z0,z1] W[z0,z1]^T with distinct strides (the PV product of the GEMM-formulated attention)
    B0, B1, M, N, K = 2, 3, 96, 64, 128
    A = _rt(B0 * B1 * M, K, seed=1)
    W = _rt(B0 * B1 * N, K, seed=2, scale=K ** -0.5)
    a_h, a_e = pair.act(A)
    w_h, w_e = pair.act(W)
    out_h = torch.zeros(B0 * B1 * M, N, dtype=torch.bfloat16, device="cuda")
    out_e = torch.zeros(B0 * B1 * M, N)
    kw = dict(M=M, N=N, batch=B0 * B1, batch_inner=B1, a_strides=(B1 * M * K, M * K), w_strides=(B1 * N * K, N * K),
              o_strides=(B1 * M * N, M * N))
    pair.hip.gemm(a_h, w_h, out_h, **kw)
    pair.emu.gemm(a_e, w_e, out_e, **kw)
    torch.cuda.synchronize()
    assert rel_l2(out_h.float().cpu(), out_e) < BF16_TOL


def test_conv_small_cin(pair):
    # the last case is large enough that a block walks several strides of its token share
    for cin, cout, (n, h, w) in ((4, 320, (2, 9, 11)), (8, 64, (2, 9, 11)), (4, 512, (2, 9, 11)), (4, 320, (4, 40, 64))):
        x = pair.act(_rt(n * h * w, cin, seed=cin))
        wt = pair.f32(_rt(cout, 9 * cin, seed=1, scale=0.2))
        b = pair.f32(_rt(cout, seed=2))
        out_h = torch.zeros(n * h * w, cout, dtype=torch.bfloat16, device="cuda")
        out_e = torch.zeros(n * h * w, cout)
        pair.run("conv_small", (x[0], n, h, w, wt[0], b[0], out_h), (x[1], n, h, w, wt[1], b[1], out_e))
        assert rel_l2(out_h.float().cpu(), out_e) < BF16_TOL


@pytest.mark.parametrize("C,c1,units,rows", [(320, 0, 16, 160), (1920, 640, 4, 70), (64, 0, 2, 1000), (2560, 1280, 2, 40),
                                             (960, 320, 1, 2560), (128, 0, 3, 4097)])
def test_groupnorm(pair, C, c1, units, rows):
    c0 = C - c1
    x0 = pair.act((_rt(units * rows, c0, seed=1) * 2.0 + 0.5).bfloat16().float())
    x1 = pair.act((_rt(units * rows, c1, seed=2) - 1.0).bfloat16().float()) if c1 else (None, None)
    gamma, beta = pair.f32(_rt(C, seed=3) * 0.1 + 1.0), pair.f32(_rt(C, seed=4) * 0.1)
    ws = torch.zeros(max(pair.hip.gn_ws_floats(units, rows), 1), device="cuda")
    st_h, st_e = torch.zeros(units, 64, device="cuda"), torch.zeros(units, 64)
    pair.run("gn_stats", (x0[0], x1[0], units, rows, 1e-5, ws, st_h), (x0[1], x1[1], units, rows, 1e-5, None, st_e))
    assert rel_l2(st_h.cpu(), st_e) < 1e-4
    for silu in (True, False):
        out_h = torch.zeros(units * rows, C, dtype=torch.bfloat16, device="cuda")
        out_e = torch.zeros(units * rows, C)
        pair.run("gn_apply", (x0[0], x1[0], units, rows, st_h, gamma[0], beta[0], silu, out_h),
                 (x0[1], x1[1], units, rows, st_e, gamma[1], beta[1], silu, out_e))
        assert rel_l2(out_h.float().cpu(), out_e) < BF16_TOL


@pytest.mark.parametrize("C,c1,units,rows", [(320, 0, 2, 77), (640, 320, 3, 40), (1280, 0, 1, 700), (2560, 1280, 2, 40),
                                              (320, 0, 1, 4000), (128, 0, 2, 9000), (960, 320, 1, 1500)])
def test_group_norm_one_call(pair, C, c1, units, rows):
    """t2v_group_norm: the workgroup-per-(group, unit) launch (1 280 x 1 x 700 on 1 024 threads, the 2 x 40 x 2 560 concat on 256), the
    fused-finish path (few slabs) and the 3-launch path (many slabs), vs the torch emulation."""
    c0 = C - c1
    x0 = pair.act((_rt(units * rows, c0, seed=1) * 2.0 + 0.5).bfloat16().float())
    x1 = pair.act((_rt(units * rows, c1, seed=2) - 1.0).bfloat16().float()) if c1 else (None, None)
    gamma, beta = pair.f32(_rt(C, seed=3) * 0.1 + 1.0), pair.f32(_rt(C, seed=4) * 0.1)
    ws = torch.zeros(pair.hip.group_norm_ws_floats(units, rows, 32, C), device="cuda")
    for silu in (True, False):
        out_h = torch.zeros(units * rows, C, dtype=torch.bfloat16, device="cuda")
        out_e = torch.zeros(units * rows, C)
        pair.run("group_norm", (x0[0], x1[0], units, rows, 1e-5, gamma[0], beta[0], silu, ws, out_h),
                 (x0[1], x1[1], units, rows, 1e-5, gamma[1], beta[1], silu, None, out_e))
        assert rel_l2(out_h.float().cpu(), out_e) < BF16_TOL
    again = torch.zeros_like(out_h)
    pair.hip.group_norm(x0[0], x1[0], units, rows, 1e-5, gamma[0], beta[0], False, ws, again)
    assert torch.equal(again, out_h)  # fixed reduction order: bit-identical on re-run


@pytest.mark.parametrize("M,C", [(1000, 320), (37, 1280), (256, 64), (5, 512), (33, 640), (9, 2048), (7, 4096)])
def test_layernorm(pair, M, C):
    x = pair.act((_rt(M, C, seed=1) * 3.0 + 1.0).bfloat16().float())
    gamma, beta = pair.f32(_rt(C, seed=3) * 0.1 + 1.0), pair.f32(_rt(C, seed=4) * 0.1)
    out_h = torch.zeros(M, C, dtype=torch.bfloat16, device="cuda")
    out_e = torch.zeros(M, C)
    pair.run("layernorm", (x[0], gamma[0], beta[0], 1e-5, out_h), (x[1], gamma[1], beta[1], 1e-5, out_e))
    assert rel_l2(out_h.float().cpu(), out_e) < BF16_TOL


@pytest.mark.parametrize("rows,n,n_pad", [(300, 2560, 2560), (64, 77, 128), (10, 40, 64)])
def test_softmax_rows(pair, rows, n, n_pad):
    s = pair.act(_rt(rows, n_pad, seed=1) * 4.0)
    pair.run("softmax_rows", (s[0], rows, n, n_pad, n_pad), (s[1], rows, n, n_pad, n_pad))
    got = s[0].float().cpu()
    assert rel_l2(got, s[1]) < 6e-3
    assert float(got[:, n:].abs().max()) == 0.0 if n_pad > n else True


@pytest.mark.parametrize("n_img,seq_q,seq_kv,heads,kv_div", [(3, 200, 200, 2, 1), (4, 160, 77, 5, 2), (2, 40, 40, 4, 1),
                                                             (2, 640, 640, 1, 1), (1, 2560, 2560, 2, 1)])
def test_attn_spatial_flash(pair, n_img, seq_q, seq_kv, heads, kv_div):
    inner = heads * 64
    n_kv = n_img // kv_div
    kp = ((seq_kv + 63) // 64) * 64
    q = pair.act(_rt(n_img * seq_q, inner, seed=1))
    k = pair.act(_rt(n_kv * seq_kv, inner, seed=2))
    vt_cpu = _rt(n_kv * inner, kp, seed=3)
    vt_cpu[:, seq_kv:] = 1e30  # padding keys: probability exactly 0, any finite V^T value must vanish
    vt = pair.act(vt_cpu)
    out_h = torch.zeros(n_img * seq_q, inner, dtype=torch.bfloat16, device="cuda")
    out_e = torch.zeros(n_img * seq_q, inner)
    scale = 0.125
    pair.run("attn_spatial", (q[0], k[0], vt[0], kp, out_h, n_img, seq_q, seq_kv, heads, kv_div, scale),
             (q[1], k[1], vt[1], kp, out_e, n_img, seq_q, seq_kv, heads, kv_div, scale))
    got = out_h.float().cpu()
    assert torch.isfinite(got).all()
    assert rel_l2(got, out_e) < 8e-3  # P is rounded to bf16 before the PV product


@pytest.mark.parametrize("n_img,seq_q,seq_kv,heads,kv_div", [(2, 700, 700, 2, 1), (4, 300, 77, 3, 2), (1, 2560, 2560, 1, 1), (2, 40, 200, 2, 1)])
def test_attn_spatial_forms_are_bit_identical(pair, n_img, seq_q, seq_kv, heads, kv_div):
    """t2v_attn_spatial_form: eight waves per workgroup (8) and 64 queries per wave with the two query sets' phases offset (65) do the SAME
    arithmetic per query as the product kernel (0) — outputs must be bit-identical: ragged query blocks of 128 / 256, the last tile's
    padding keys, per-clip text keys (kv_div), a running max that moves.  (Both are measured no faster: tools / tests only.)"""
    from t2v_turbo_amd import native as _nt
    if not _nt.has_experimental():
        pytest.skip("entry point of a T2V_EXPERIMENTAL=1 build (python t2v-turbo_amd/csrc/build.py with T2V_EXPERIMENTAL=1, then T2V_HIP_LIB=.../libt2v_hip_exp.so)")

    inner = heads * 64
    n_kv = n_img // kv_div
    kp = ((seq_kv + 63) // 64) * 64
    q = _rt(n_img * seq_q, inner, seed=11, scale=2.0).cuda().bfloat16()
    k = _rt(n_kv * seq_kv, inner, seed=12, scale=2.0).cuda().bfloat16()
    vt = _rt(n_kv * inner, kp, seed=13).cuda().bfloat16()
    vt[:, seq_kv:] = 1e30
    outs = []
    try:
        for form in (0, 8, 65):
            pair.hip.lib.t2v_attn_spatial_form(form)
            o = torch.full((n_img * seq_q, inner), float("nan"), dtype=torch.bfloat16, device="cuda")
            pair.hip.attn_spatial(q, k, vt, kp, o, n_img, seq_q, seq_kv, heads, kv_div, 0.125)
            torch.cuda.synchronize()
            outs.append(o.clone())
    finally:
        pair.hip.lib.t2v_attn_spatial_form(0)
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_attn_spatial_online_softmax_rescale(pair):
    """Force the running max to jump at a late key tile (spike one key against every query)."""
    seq, inner = 256, 64
    qc = _rt(seq, inner, seed=1)
    kc = _rt(seq, inner, seed=2)
    kc[200] = qc.mean(dim=0) * 0 + 6.0 * torch.sign(qc[0])  # large dot with many queries, in the 4th tile
    q, k = pair.act(qc), pair.act(kc)
    vt = pair.act(_rt(inner, seq, seed=3))
    out_h = torch.zeros(seq, inner, dtype=torch.bfloat16, device="cuda")
    out_e = torch.zeros(seq, inner)
    pair.run("attn_spatial", (q[0], k[0], vt[0], seq, out_h, 1, seq, seq, 1, 1, 0.125),
             (q[1], k[1], vt[1], seq, out_e, 1, seq, seq, 1, 1, 0.125))
    assert rel_l2(out_h.float().cpu(), out_e) < 8e-3


@pytest.mark.parametrize("clips,frames,hw,heads", [(1, 16, 100, 5), (2, 4, 64, 2), (1, 8, 33, 1), (1, 24, 16, 2), (2, 40, 9, 3), (1, 1, 50, 2)])
def test_attn_temporal(pair, clips, frames, hw, heads):
    inner = heads * 64
    M = clips * frames * hw
    qkv = pair.act(_rt(M, 3 * inner, seed=1))
    out_h = torch.zeros(M, inner, dtype=torch.bfloat16, device="cuda")
    out_e = torch.zeros(M, inner)
    pr_h = torch.zeros(clips * hw * heads, frames, frames, device="cuda")
    pr_e = torch.zeros(clips * hw * heads, frames, frames)
    sl = lambda t: (t[:, :inner], t[:, inner:2 * inner], t[:, 2 * inner:])
    pair.run("attn_temporal", (*sl(qkv[0]), out_h, clips, frames, hw, heads, 0.125, pr_h),
             (*sl(qkv[1]), out_e, clips, frames, hw, heads, 0.125, pr_e))
    assert rel_l2(out_h.float().cpu(), out_e) < BF16_TOL
    assert rel_l2(pr_h.cpu(), pr_e) < 1e-4


def test_layout_embedding_elementwise(pair):
    x = _rt(2, 4, 3, 5, 7, seed=1)
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        tok = torch.zeros(2 * 3 * 35, 4, dtype=torch.bfloat16, device="cuda")
        pair.hip.ncfhw_to_tokens(x.to(dt).cuda(), tok)
        ref = torch.zeros(2 * 3 * 35, 4)
        pair.emu.ncfhw_to_tokens(x, ref)
        assert torch.equal(tok.float().cpu(), ref)
        back = torch.zeros(2, 4, 3, 5, 7, dtype=dt, device="cuda")
        pair.hip.tokens_to_ncfhw(tok, back)
        assert torch.equal(back.float().cpu(), x)
    ts = torch.tensor([999, 0, 519], dtype=torch.int64)
    e_h = torch.zeros(3, 320, dtype=torch.bfloat16, device="cuda")
    e_e = torch.zeros(3, 320)
    pair.run("timestep_embedding", (ts.cuda(), 320, False, e_h), (ts, 320, False, e_e))
    assert (e_h.float().cpu() - e_e).abs().max() < 1.2e-2  # bf16 rounding + fp32 sincos at |arg| <= 999
    w = torch.tensor([7.5, 12.25])
    g_h = torch.zeros(2, 256, dtype=torch.bfloat16, device="cuda")
    g_e = torch.zeros(2, 256)
    pair.run("timestep_embedding", (w.cuda(), 256, True, g_h), (w, 256, True, g_e))
    assert (g_h.float().cpu() - g_e).abs().max() < 2e-2
    v = _rt(1000, seed=2)
    s_h = torch.zeros(1000, dtype=torch.bfloat16, device="cuda")
    pair.hip.silu(v.cuda().bfloat16(), s_h)
    assert rel_l2(s_h.float().cpu(), torch.nn.functional.silu(v)) < BF16_TOL
    # scheduler family
    xs, eps, noise = _rt(2, 4, 3, 5, 5, seed=3), _rt(2, 4, 3, 5, 5, seed=4), _rt(2, 4, 3, 5, 5, seed=5)
    prev_h, den_h = torch.zeros_like(xs).cuda(), torch.zeros_like(xs).cuda()
    prev_e, den_e = torch.zeros_like(xs), torch.zeros_like(xs)
    args = (0.3, 0.95, 0.01, 0.99, 0.6, 0.8)
    pair.hip.lcm_step(xs.cuda(), eps.cuda(), noise.cuda(), *args, prev_h, den_h)
    pair.emu.lcm_step(xs, eps, noise, *args, prev_e, den_e)
    assert rel_l2(prev_h.cpu(), prev_e) < 1e-6 and rel_l2(den_h.cpu(), den_e) < 1e-6
    o_h, o_e = torch.zeros_like(xs).cuda(), torch.zeros_like(xs)
    pair.hip.lincomb3(xs.cuda(), eps.cuda(), None, [0.5, 2.0], [1.5, -1.0], None, o_h)
    pair.emu.lincomb3(xs, eps, None, [0.5, 2.0], [1.5, -1.0], None, o_e)
    assert rel_l2(o_h.cpu(), o_e) < 1e-6


def test_bad_arguments_return_errors(pair):
    from t2v_turbo_amd.native import NativeError
    a = torch.zeros(64, 48, dtype=torch.bfloat16, device="cuda")  # channels not a multiple of 64
    w = torch.zeros(64, 48, dtype=torch.bfloat16, device="cuda")
    out = torch.zeros(64, 64, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(NativeError):
        pair.hip.gemm(a, w, out, M=64, N=64)
    with pytest.raises(NativeError):
        pair.hip.attn_temporal(out, out, out, out, 1, 2000, 1, 1, 0.125)


# ------------------------------------------------------------------------------------ backward pieces (VAE decoder dX)
@pytest.mark.parametrize("C,units,rows,silu,with_resid", [(128, 2, 300, True, False), (512, 1, 257, True, True),
                                                          (256, 3, 64, False, True), (64, 2, 5000, True, False)])
def test_gn_bwd(pair, C, units, rows, silu, with_resid):
    M = units * rows
    x = pair.act((_rt(M, C, seed=1) * 1.5 + 0.3).bfloat16().float())
    dy = pair.act((_rt(M, C, seed=2)).bfloat16().float())
    rs = pair.act((_rt(M, C, seed=5)).bfloat16().float()) if with_resid else (None, None)
    gamma, beta = pair.f32(_rt(C, seed=3) * 0.1 + 1.0), pair.f32(_rt(C, seed=4) * 0.1)
    st_h, st_e = torch.zeros(units, 64, device="cuda"), torch.zeros(units, 64)
    ws = torch.zeros(max(pair.hip.gn_ws_floats(units, rows), 1), device="cuda")
    pair.run("gn_stats", (x[0], None, units, rows, 1e-6, ws, st_h), (x[1], None, units, rows, 1e-6, None, st_e))
    wsb = torch.zeros(pair.hip.gn_bwd_ws_floats(units, rows), device="cuda")
    dx_h = torch.zeros(M, C, dtype=torch.bfloat16, device="cuda")
    dx_e = torch.zeros(M, C)
    pair.run("gn_bwd", (x[0], units, rows, st_h, gamma[0], beta[0], silu, dy[0], rs[0], wsb, dx_h),
             (x[1], units, rows, st_e, gamma[1], beta[1], silu, dy[1], rs[1], None, dx_e))
    assert rel_l2(dx_h.float().cpu(), dx_e) < BF16_TOL
    # and the emulation itself against torch autograd of group_norm (+ silu)
    xa = x[1].clone().requires_grad_(True)
    y = torch.nn.functional.group_norm(xa.reshape(units, rows, C).transpose(1, 2), 32, gamma[1], beta[1], 1e-6)
    y = torch.nn.functional.silu(y) if silu else y
    (y.transpose(1, 2).reshape(M, C) * dy[1]).sum().backward()
    ref = xa.grad + (rs[1] if with_resid else 0)
    assert rel_l2(dx_e, ref) < 1e-4


@pytest.mark.parametrize("rows,n,n_pad", [(300, 2560, 2560), (64, 77, 128), (10, 40, 64)])
def test_softmax_bwd_rows(pair, rows, n, n_pad):
    logits = _rt(rows, n_pad, seed=1) * 3.0
    p = torch.zeros(rows, n_pad)
    p[:, :n] = logits[:, :n].softmax(dim=1)
    pp = pair.act(p.bfloat16().float())
    dp = pair.act(_rt(rows, n_pad, seed=2).bfloat16().float())
    pair.run("softmax_bwd_rows", (pp[0], dp[0], rows, n, n_pad, n_pad), (pp[1], dp[1], rows, n, n_pad, n_pad))
    assert rel_l2(dp[0].float().cpu(), dp[1]) < BF16_TOL
    assert float(dp[0][:, n:].float().abs().max()) == 0.0 if n_pad > n else True


def test_transpose_and_sumpool(pair):
    src = pair.act(_rt(3 * 70, 200, seed=1).bfloat16().float())
    out_h = torch.zeros(3 * 200, 72, dtype=torch.bfloat16, device="cuda")
    out_e = torch.zeros(3 * 200, 72)
    pair.run("transpose", (src[0], 70, 200, out_h), (src[1], 70, 200, out_e), dict(batch=3, in_stride=70 * 200, out_stride=200 * 72),
             dict(batch=3, in_stride=70 * 200, out_stride=200 * 72))
    assert torch.equal(out_h.float().cpu(), out_e)
    x = pair.act(_rt(2 * 6 * 10, 64, seed=2).bfloat16().float())
    o_h = torch.zeros(2 * 3 * 5, 64, dtype=torch.bfloat16, device="cuda")
    o_e = torch.zeros(2 * 3 * 5, 64)
    pair.run("sumpool2x2", (x[0], 2, 3, 5, o_h), (x[1], 2, 3, 5, o_e))
    assert rel_l2(o_h.float().cpu(), o_e) < BF16_TOL


@pytest.mark.parametrize("cfg", [0, 4, 12, 19])
def test_gemm_fast_and_generic_epilogues_agree_with_emulation(pair, cfg):
    """Both instantiations of every tile: the fast one (accumulators start at bias + row vector + residual; taken when
    operands are vector aligned, N % 16 == 0 and alpha == 1) and the generic one (alpha != 1, N % 16 != 0, fp32 output)."""
    fast = dict(M=700, N=320, c0=320, residual=True, rowvec_div=70, cfg=cfg)
    assert _gemm_case(pair, **fast) < BF16_TOL
    assert _gemm_case(pair, **{**fast, "act": 2}) < BF16_TOL                        # SiLU after the folded terms
    assert _gemm_case(pair, **{**fast, "alpha": 0.37}) < BF16_TOL                   # alpha != 1 -> generic
    assert _gemm_case(pair, M=700, N=72, c0=128, residual=True, cfg=cfg) < BF16_TOL  # N % 16 != 0 -> generic
    assert _gemm_case(pair, M=333, N=320, c0=64, out_f32=True, rowvec_div=333, cfg=cfg) < 1e-4  # fp32 out
    n, h, w = 4, 10, 16
    assert _gemm_case(pair, M=n * h * w, N=128, c0=64, c1=64, mode=1, n_img=n, h=h, w=w, rows=n * h * w, residual=True,
                      rowvec_div=h * w, cfg=cfg, seed=3) < BF16_TOL                  # conv + concat + both folded terms


@pytest.mark.parametrize("offset,scale", [(8.0, 0.5), (-30.0, 1.0), (3.0, 0.05)])
@pytest.mark.parametrize("C,c1,units,rows", [(320, 0, 2, 77), (960, 320, 1, 1500), (320, 0, 1, 9000)])
def test_group_norm_inputs_with_mean_far_from_zero(pair, C, c1, units, rows, offset, scale):
    """Real VideoCrafter2 activations are not zero-mean: |mean| >> std must not cost the E[x^2] - mean^2 statistics their
    digits (fp32 partial sums).  The emulation works on the same bf16-rounded inputs, so the tolerance stays the bf16 one."""
    c0 = C - c1
    x0 = pair.act((_rt(units * rows, c0, seed=1) * scale + offset).bfloat16().float())
    x1 = pair.act((_rt(units * rows, c1, seed=2) * scale - offset).bfloat16().float()) if c1 else (None, None)
    gamma, beta = pair.f32(_rt(C, seed=3) * 0.1 + 1.0), pair.f32(_rt(C, seed=4) * 0.1)
    ws = torch.zeros(pair.hip.group_norm_ws_floats(units, rows, 32, C), device="cuda")
    out_h = torch.zeros(units * rows, C, dtype=torch.bfloat16, device="cuda")
    out_e = torch.zeros(units * rows, C)
    pair.run("group_norm", (x0[0], x1[0], units, rows, 1e-5, gamma[0], beta[0], False, ws, out_h),
             (x0[1], x1[1], units, rows, 1e-5, gamma[1], beta[1], False, None, out_e))
    assert rel_l2(out_h.float().cpu(), out_e) < BF16_TOL


@pytest.mark.parametrize("offset,scale", [(8.0, 0.5), (-30.0, 1.0), (3.0, 0.05)])
@pytest.mark.parametrize("M,C", [(1000, 320), (37, 1280)])
def test_layernorm_inputs_with_mean_far_from_zero(pair, M, C, offset, scale):
    x = pair.act((_rt(M, C, seed=1) * scale + offset).bfloat16().float())
    gamma, beta = pair.f32(_rt(C, seed=3) * 0.1 + 1.0), pair.f32(_rt(C, seed=4) * 0.1)
    out_h = torch.zeros(M, C, dtype=torch.bfloat16, device="cuda")
    out_e = torch.zeros(M, C)
    pair.run("layernorm", (x[0], gamma[0], beta[0], 1e-5, out_h), (x[1], gamma[1], beta[1], 1e-5, out_e))
    assert rel_l2(out_h.float().cpu(), out_e) < BF16_TOL


def test_fill_zero_edges(pair):
    """Zero fill is a kernel (a memset node of a captured launch list did not re-execute in order on replay): unaligned head
    and tail bytes, nothing outside the range touched."""
    buf = torch.full((4096 + 64,), 0x5A, dtype=torch.uint8, device="cuda")
    for off, n in ((0, 4096), (3, 1000), (17, 31), (5, 1), (16, 16), (1, 4095)):
        buf.fill_(0x5A)
        pair.hip._call("t2v_fill_zero", buf.data_ptr() + off, n)
        torch.cuda.synchronize()
        got = buf.cpu()
        assert int(got[off:off + n].max()) == 0
        assert bool((got[:off] == 0x5A).all()) and bool((got[off + n:] == 0x5A).all())


@pytest.mark.parametrize("C,c1,units,rows", [(320, 0, 1, 40960), (640, 0, 1, 10240), (1280, 0, 1, 2560), (1280, 0, 1, 640), (320, 0, 16, 2560),
                                             (640, 320, 16, 640), (2560, 1280, 16, 40), (960, 320, 16, 2560), (64, 0, 3, 50)])
def test_group_norm_one_launch_form_at_unet_sizes(pair, C, c1, units, rows):
    """The one-launch GroupNorm (registers hold the tensor, per-unit inter-workgroup barrier) at the sizes the UNet calls it with,
    against the three-launch form of the same entry point and the emulation; 200 back-to-back calls on alternating inputs (a stale
    partial or a missed barrier generation shows as a wrong result), bit-identical on re-run, no barrier timeout recorded.
    (Round 5: shapes whose groups are a multiple of 8 channels wide with at most 4 096 chunks per (group, unit) — 1 280 x 1 x 640 and
    2 560 x 16 x 40 here — take the workgroup-per-group launch (gn_group_kernel) in BOTH modes: it needs no barrier and comes first.)"""
    lib = pair.hip.lib
    c0 = C - c1
    xs = []
    for sd, off in ((1, 0.5), (11, -2.0)):
        x0 = pair.act((_rt(units * rows, c0, seed=sd) * 2.0 + off).bfloat16().float())
        x1 = pair.act((_rt(units * rows, c1, seed=sd + 1) - off).bfloat16().float()) if c1 else (None, None)
        xs.append((x0, x1))
    gamma, beta = pair.f32(_rt(C, seed=3) * 0.1 + 1.0), pair.f32(_rt(C, seed=4) * 0.1)
    ws = torch.zeros(pair.hip.group_norm_ws_floats(units, rows, 32, C), device="cuda")
    outs = {}
    try:
        for mode in (1, 0):
            lib.t2v_gn_coop_enable(mode)
            for k, (x0, x1) in enumerate(xs):
                o = torch.zeros(units * rows, C, dtype=torch.bfloat16, device="cuda")
                pair.hip.group_norm(x0[0], x1[0], units, rows, 1e-5, gamma[0], beta[0], True, ws, o)
                outs[(mode, k)] = o
        torch.cuda.synchronize()
        lib.t2v_gn_coop_enable(1)
        o2 = [torch.zeros_like(outs[(1, 0)]) for _ in range(2)]
        for it in range(200):
            k = it & 1
            pair.hip.group_norm(xs[k][0][0], xs[k][1][0], units, rows, 1e-5, gamma[0], beta[0], True, ws, o2[k])
        torch.cuda.synchronize()
    finally:
        lib.t2v_gn_coop_enable(0)   # the library default (the one-launch form is opt-in: measured slower, norm.hip)
    assert lib.t2v_gn_coop_error() == 0
    for k, (x0, x1) in enumerate(xs):
        ref = torch.zeros(units * rows, C)
        pair.emu.group_norm(x0[1], x1[1], units, rows, 1e-5, gamma[1], beta[1], True, None, ref)
        assert rel_l2(outs[(1, k)].float().cpu(), ref) < BF16_TOL
        assert rel_l2(outs[(0, k)].float().cpu(), ref) < BF16_TOL
        assert torch.equal(o2[k], outs[(1, k)])


@pytest.mark.parametrize("M,K,res", [(40960, 320, True), (320, 320, False), (1000, 1280, True)])
def test_gemm_layernorm_second_output(pair, M, K, res):
    """t2v_gemm ln_* fields on the 160x320 tile: main output bit-identical to the plain launch of the same tile, LayerNorm output
    against the emulation (LN of the fp32 epilogue values) and against t2v_layernorm of the bf16 main output; residual stream with
    |mean| >> std; M not a multiple of the tile."""
    N = 320
    a = pair.act(_rt(M, K, seed=1))
    wt = pair.act(_rt(N, K, seed=2, scale=K ** -0.5))
    b = pair.f32(_rt(N, seed=3))
    r = pair.act((_rt(M, N, seed=5) * 0.5 + 6.0).bfloat16().float()) if res else (None, None)
    gamma, beta = pair.f32(_rt(N, seed=6) * 0.1 + 1.0), pair.f32(_rt(N, seed=7) * 0.1)
    out_h = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    ln_h = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    out_p = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    ln_p = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    out_e, ln_e = torch.zeros(M, N), torch.zeros(M, N)
    pair.hip.gemm(a[0], wt[0], out_h, M=M, N=N, bias=b[0], residual=r[0], ln=(gamma[0], beta[0], 1e-5, ln_h))
    pair.hip.gemm(a[0], wt[0], out_p, M=M, N=N, bias=b[0], residual=r[0], tile_cfg=23, split_k=1)
    pair.hip.layernorm(out_p, gamma[0], beta[0], 1e-5, ln_p)
    pair.emu.gemm(a[1], wt[1], out_e, M=M, N=N, bias=b[1], residual=r[1], ln=(gamma[1], beta[1], 1e-5, ln_e))
    torch.cuda.synchronize()
    assert torch.equal(out_h, out_p)
    assert torch.isfinite(ln_h.float()).all()
    assert rel_l2(ln_h.float().cpu(), ln_e) < BF16_TOL
    assert rel_l2(ln_h.float().cpu(), ln_p.float().cpu()) < 2e-2   # LN of the bf16-rounded stream: the rounding of x shows when |mean| >> std


def test_conv3x3_small_cout_direct(pair):
    """t2v_conv3x3_small_cout at the VAE decoder's conv_out size (one 320x512 frame, 128 -> 3 channels, fp32 out) and small corner cases."""
    from t2v_turbo_amd import native as _nt
    if not _nt.has_experimental():
        pytest.skip("entry point of a T2V_EXPERIMENTAL=1 build (python t2v-turbo_amd/csrc/build.py with T2V_EXPERIMENTAL=1, then T2V_HIP_LIB=.../libt2v_hip_exp.so)")

    for n_img, h, w, cin, cout, f32 in [(1, 320, 512, 128, 3, True), (2, 6, 8, 16, 3, False), (1, 5, 12, 64, 4, True), (3, 3, 4, 8, 1, True)]:
        M = n_img * h * w
        x = _rt(M, cin, seed=cin + h)
        wgt = _rt(cout, 9 * cin, seed=3, scale=(9 * cin) ** -0.5)
        bias = _rt(cout, seed=4)
        assert pair.hip.conv_small_cout_supported(w, cin, cout)
        o_h = torch.full((M, cout), float("nan"), device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
        o_e = torch.zeros(M, cout)
        pair.hip.conv_small_cout(x.cuda().bfloat16().contiguous(), n_img, h, w, wgt.cuda().float().contiguous(), bias.cuda().float().contiguous(), o_h)
        pair.emu.conv_small(x, n_img, h, w, wgt, bias, o_e)
        torch.cuda.synchronize()
        assert torch.isfinite(o_h.float()).all()
        assert rel_l2(o_h.float().cpu(), o_e) < (2e-5 if f32 else BF16_TOL), (n_img, h, w, cin, cout)


# ---------------------------------------------------------------------------------- t2v_conv_halo (csrc/conv_halo.hip)
def _halo_case(pair, *, n_img, h, w, c0, N, c1=0, cfg=0, rowvec=False, residual=False, act=0, colstat=False, seed=0, repeat=1, ups=0):
    """3x3 conv on the halo-slab kernel (slab-major pack) against the emulated conv on the same bf16-rounded data; with a residual
    both add it to the bf16-rounded product (the kernel's row pass; tests/emu_ops.py::conv_halo).  ``ups`` = 1: nearest-x2 upsampled
    source (T2V_GEMM_CONV3X3_UP2; (h, w) is the source grid)."""
    from t2v_turbo_amd import native as nt
    m_src = n_img * h * w
    M, K = m_src << (2 * ups), 9 * (c0 + c1)
    a0 = _rt(m_src, c0, seed=seed)
    a1 = _rt(m_src, c1, seed=seed + 1) if c1 else None
    wt = nt.pack_conv_slab(_rt(N, K, seed=seed + 2, scale=K ** -0.5))
    b, rv, res = _rt(N, seed=seed + 3), (_rt(n_img, N, seed=seed + 4) if rowvec else None), (_rt(M, N, seed=seed + 5) if residual else None)
    outs, stats = [], []
    for side, ops in enumerate((pair.hip, pair.emu)):
        cvt = (lambda t: None if t is None else t.cuda().bfloat16().contiguous()) if side == 0 else (lambda t: None if t is None else t.clone())
        f32 = (lambda t: None if t is None else t.cuda().float().contiguous()) if side == 0 else (lambda t: None if t is None else t.clone())
        dev = "cuda" if side == 0 else "cpu"
        out = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16 if side == 0 else torch.float32)
        cs = torch.full((M // 32, N, 2), float("nan"), device=dev) if colstat else None
        kw = dict(M=M, N=N, a1=cvt(a1), mode=nt.GEMM_CONV3X3_UP2 if ups else nt.GEMM_CONV3X3, n_img=n_img, h=h, wd=w, bias=f32(b), rowvec=f32(rv),
                  rowvec_div=((h * w) << (2 * ups)) if rowvec else 0,
                  residual=cvt(res), act=act, tile_cfg=cfg)
        if colstat:
            kw["colstat"] = cs
        x0, ww = cvt(a0), cvt(wt)
        assert ops.conv_halo_supported(x0, ww, out, **kw) == 1
        for _ in range(repeat if side == 0 else 1):
            ops.conv_halo(x0, ww, out, **kw)
        if side == 0:
            torch.cuda.synchronize()
        outs.append(out.float().cpu())
        stats.append(None if cs is None else cs.cpu())
    y, r = outs
    assert torch.isfinite(y).all()
    assert rel_l2(y, r) < BF16_TOL, rel_l2(y, r)
    if colstat:   # of the bf16 values the kernel itself stored
        yo = y.reshape(M // 32, 32, N)
        want = torch.stack([yo.sum(dim=1), (yo * yo).sum(dim=1)], dim=2)
        assert torch.allclose(stats[0], want, rtol=1e-4, atol=2e-3), (stats[0] - want).abs().max()


@pytest.mark.parametrize("cfg", [40, 41, 43])
def test_conv_halo_tiles_32_wide(pair, cfg):
    # 20x32 grid, 320 -> 160 channels: 10 sub-slabs = 45 / 45 / 23 weight stages: the DMA ring wraps many times, the loaders run
    # ahead past the end; residual row pass, row vector, column statistics
    _halo_case(pair, n_img=4, h=20, w=32, c0=320, N=160, cfg=cfg, rowvec=True, residual=True, colstat=True, seed=cfg)


def test_conv_halo_16_wide_whole_frames(pair):
    _halo_case(pair, n_img=6, h=10, w=16, c0=640, N=160, cfg=42, rowvec=True, residual=True, colstat=True, seed=7)


def test_conv_halo_virtual_concat_ragged_and_silu(pair):
    _halo_case(pair, n_img=2, h=12, w=64, c0=128, c1=192, N=80, cfg=41, act=2, seed=8)      # ragged rows, two tile columns
    _halo_case(pair, n_img=2, h=20, w=32, c0=64, c1=256, N=48, cfg=40, residual=True, seed=9)   # ragged channel tile


def test_conv_halo_vae_decoder_widths(pair):
    # 128 / 256 / 512 output channels (KL-VAE decoder, ae_modules.py:183-203) on padded 80-channel wave tiles, at decoder image sizes
    _halo_case(pair, n_img=2, h=80, w=128, c0=128, N=128, residual=True, colstat=True, seed=21, repeat=2)
    _halo_case(pair, n_img=2, h=40, w=64, c0=256, N=256, colstat=True, seed=22)
    _halo_case(pair, n_img=2, h=40, w=64, c0=512, N=512, residual=True, colstat=True, seed=23)
    _halo_case(pair, n_img=1, h=80, w=128, c0=512, N=256, colstat=True, seed=24)


def test_conv_halo_64_channel_wave_tiles(pair):
    # tile id 44 (320x128 on 80 x 64 wave tiles) forced: exact widths, a ragged one (N = 192), row vector + SiLU, virtual concat
    _halo_case(pair, n_img=2, h=40, w=64, c0=128, N=128, cfg=44, rowvec=True, residual=True, colstat=True, seed=31, repeat=3)
    _halo_case(pair, n_img=2, h=20, w=32, c0=256, c1=256, N=512, cfg=44, colstat=True, seed=32)
    _halo_case(pair, n_img=1, h=12, w=64, c0=64, N=192, cfg=44, rowvec=True, act=2, seed=33)


def test_conv_halo_over_nearest_x2_upsampled_source(pair):
    # the Upsample convs (interpolate x2 + 3x3 conv) of the UNet's decoder half at full size (one frame pair) and of the VAE decoder
    _halo_case(pair, n_img=2, h=20, w=32, c0=640, N=640, colstat=True, seed=41, ups=1, repeat=2)       # -> 40x64
    _halo_case(pair, n_img=2, h=10, w=16, c0=1280, N=1280, colstat=True, seed=42, ups=1)               # -> 20x32
    _halo_case(pair, n_img=2, h=5, w=8, c0=1280, N=1280, colstat=True, seed=43, ups=1)                 # -> 10x16 (whole-frame tiles)
    _halo_case(pair, n_img=1, h=80, w=128, c0=256, N=256, colstat=True, seed=44, ups=1)                # VAE: -> 160x256
    _halo_case(pair, n_img=2, h=6, w=16, c0=64, c1=64, N=80, cfg=41, rowvec=True, residual=True, act=2, seed=45, ups=1)   # ragged rows


def test_conv_halo_unet_level_shapes_repeatable(pair):
    # the three levels at full size (one frame pair each), library's own tile choice; 3 back-to-back launches into the same output
    _halo_case(pair, n_img=2, h=40, w=64, c0=320, N=320, rowvec=True, residual=True, colstat=True, seed=10, repeat=3)
    _halo_case(pair, n_img=2, h=20, w=32, c0=1280, c1=640, N=640, colstat=True, seed=11, repeat=3)
    _halo_case(pair, n_img=2, h=10, w=16, c0=1280, N=1280, rowvec=True, residual=True, colstat=True, seed=12, repeat=3)


# ---- t2v_linear_pr: short-K Linear with the activation panel resident in LDS (csrc/linear_pr.hip) ------------------------------------
def _lpr_case(pair, *, M, K, N, act=0, bias=True, residual=False, ny=0, seed=0, repeat=1, lda=None, ldo=None, ln_in=False):
    """t2v_linear_pr on the fragment pack against the emulated Linear on the same bf16-rounded data (tests/emu_ops.py::linear_pr);
    ``repeat`` back-to-back launches into the same output (the weight ring and the residual prefetch run ahead of the stores)."""
    from t2v_turbo_amd import native as nt
    n_out = N // 2 if act == nt.ACT_GEGLU else N
    a = _rt(M, lda or K, seed=seed)
    if ln_in:   # rows with their own offsets and scales
        a = (a * (0.5 + torch.arange(M)[:, None] % 7) + (torch.arange(M)[:, None] % 5 - 2.0)).bfloat16().float()
    w = _rt(N, K, seed=seed + 1, scale=K ** -0.5)
    b = _rt(N, seed=seed + 2) if bias else None
    res = _rt(M, n_out, seed=seed + 3) if residual else None
    ln = (1.0 + 0.2 * _rt(K, seed=seed + 4), 0.3 * _rt(K, seed=seed + 5), 1e-5) if ln_in else None
    outs = []
    pair.hip.lib.t2v_linear_pr_force_split(ny)
    try:
        for side, ops in enumerate((pair.hip, pair.emu)):
            cvt = (lambda t: None if t is None else t.cuda().bfloat16().contiguous()) if side == 0 else (lambda t: None if t is None else t.clone())
            f32 = (lambda t: None if t is None else t.cuda().float().contiguous()) if side == 0 else (lambda t: None if t is None else t.clone())
            dev = "cuda" if side == 0 else "cpu"
            out_full = torch.full((M, ldo or n_out), float("nan"), device=dev, dtype=torch.bfloat16 if side == 0 else torch.float32)
            out = out_full[:, :n_out]
            wp = cvt(nt.pack_linear_pr(w.bfloat16()).float())
            kw = dict(M=M, N=N, bias=f32(b), residual=cvt(res), act=act)
            if ln_in:
                kw["ln_in"] = (f32(ln[0]), f32(ln[1]), ln[2])
            x = cvt(a)[:, :K]
            assert ops.linear_pr_supported(x, wp, out, **kw) == 1
            for _ in range(repeat if side == 0 else 1):
                ops.linear_pr(x, wp, out, **kw)
            if side == 0:
                torch.cuda.synchronize()
            outs.append(out.float().cpu())
            if ldo:
                assert torch.isnan(out_full[:, n_out:].float()).all()
    finally:
        pair.hip.lib.t2v_linear_pr_force_split(0)
    y, r = outs
    assert torch.isfinite(y).all()
    assert rel_l2(y, r) < BF16_TOL, rel_l2(y, r)
    assert (y - r).abs().max() < 0.05 * r.abs().max()


def test_linear_pr_unet_shapes_repeatable(pair):
    # the launches the engine routes here, at full size: GEGLU projections and q | k | v / q | k of the 320- and 640-channel levels
    from t2v_turbo_amd import native as nt
    _lpr_case(pair, M=40960, K=320, N=2560, act=nt.ACT_GEGLU, seed=1, repeat=3)
    _lpr_case(pair, M=40960, K=320, N=960, bias=False, seed=2, repeat=2)
    _lpr_case(pair, M=40960, K=320, N=640, bias=False, seed=3)
    _lpr_case(pair, M=10240, K=640, N=5120, act=nt.ACT_GEGLU, seed=4, repeat=3)
    _lpr_case(pair, M=10240, K=640, N=1920, bias=False, seed=5, repeat=2)
    _lpr_case(pair, M=40960, K=512, N=4096, act=nt.ACT_GEGLU, ln_in=True, seed=14, repeat=2)   # init_attn: K = 512 on 96-row panels
    _lpr_case(pair, M=40960, K=512, N=1536, bias=False, ln_in=True, seed=15)


def test_linear_pr_residual_and_few_chunks(pair):
    # N = C with bias + residual (five / ten chunks: idle waves, the residual tile as the first step's C operand), without either
    _lpr_case(pair, M=40960, K=320, N=320, residual=True, seed=6, repeat=2)
    _lpr_case(pair, M=10240, K=640, N=640, residual=True, seed=7, repeat=2)
    _lpr_case(pair, M=2560, K=320, N=320, bias=False, seed=8)
    _lpr_case(pair, M=10240, K=640, N=1280, residual=True, ny=3, seed=9)


def test_linear_pr_layernorm_in_the_panel_fill(pair):
    # ln_in (t2v_gemm_desc): LayerNorm(x) -> q | k | v and -> GEGLU projection as one launch, at the UNet's sizes and on ragged / strided ones
    from t2v_turbo_amd import native as nt
    _lpr_case(pair, M=40960, K=320, N=2560, act=nt.ACT_GEGLU, ln_in=True, seed=21, repeat=2)
    _lpr_case(pair, M=40960, K=320, N=960, bias=False, ln_in=True, seed=22)
    _lpr_case(pair, M=10240, K=640, N=5120, act=nt.ACT_GEGLU, ln_in=True, seed=23, repeat=2)
    _lpr_case(pair, M=10240, K=640, N=1920, bias=False, ln_in=True, seed=24)
    _lpr_case(pair, M=1000, K=320, N=1280, act=nt.ACT_GEGLU, ln_in=True, seed=25)
    _lpr_case(pair, M=333, K=640, N=1024, ln_in=True, ny=2, seed=26, lda=704, ldo=1152)


def test_gn_coef_cs_and_groupnorm_in_the_panel_fill(pair):
    """t2v_gn_coef_cs + t2v_linear_pr with gn_coef (x = proj_in(norm(x)) of the transformers as statistics launch + ONE GEMM launch) against
    the emulated pair, at the 320-channel level's two unit layouts (16 frames of 2 560 rows; one clip of 40 960 rows)."""
    from t2v_turbo_amd import native as nt
    for units, rpu, seed in ((16, 2560, 51), (1, 40960, 52)):
        M, K, N, G = units * rpu, 320, 320, 32
        x = _rt(M, K, seed=seed) * (1.0 + (torch.arange(M)[:, None] // rpu) % 3) + 0.3
        x = x.bfloat16().float()
        w, b = _rt(N, K, seed=seed + 1, scale=K ** -0.5), _rt(N, seed=seed + 2)
        gamma, beta = 1.0 + 0.2 * _rt(K, seed=seed + 3), 0.3 * _rt(K, seed=seed + 4)
        cs = torch.stack([x.view(M // 32, 32, K).sum(dim=1), (x * x).view(M // 32, 32, K).sum(dim=1)], dim=2).contiguous()   # [M/32, K, 2]
        outs = []
        for side, ops in enumerate((pair.hip, pair.emu)):
            dev = "cuda" if side == 0 else "cpu"
            cvt = (lambda t: t.cuda().bfloat16().contiguous()) if side == 0 else (lambda t: t.clone())
            f32 = lambda t: t.to(dev).float().contiguous()   # noqa: E731
            coef = torch.full((units, 2 * K), float("nan"), device=dev)
            assert ops.gn_coef_cs_supported(f32(cs), None, K, 0, units, rpu, G)
            ops.gn_coef_cs(f32(cs), None, K, 0, units, rpu, 1e-6, f32(gamma), f32(beta), coef, G)
            out = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16 if side == 0 else torch.float32)
            wp = cvt(nt.pack_linear_pr(w.bfloat16()).float())
            kw = dict(M=M, N=N, bias=f32(b), gn_in=(coef, rpu))
            assert ops.linear_pr_supported(cvt(x), wp, out, **kw) == 1
            ops.linear_pr(cvt(x), wp, out, **kw)
            outs.append((coef.float().cpu(), out.float().cpu()))
        torch.cuda.synchronize()
        assert rel_l2(outs[0][0], outs[1][0]) < 1e-4
        assert rel_l2(outs[0][1], outs[1][1]) < BF16_TOL


def test_linear_pr_ragged_rows_strides_and_splits(pair):
    from t2v_turbo_amd import native as nt
    _lpr_case(pair, M=1000, K=320, N=1280, act=nt.ACT_GEGLU, seed=10)                 # ragged last panel
    _lpr_case(pair, M=333, K=640, N=1024, act=nt.ACT_GEGLU, ny=2, seed=11)            # ragged, two workgroup rows
    _lpr_case(pair, M=4000, K=320, N=960, seed=12, lda=384, ldo=1024)                 # operands as column slices of wider buffers
    _lpr_case(pair, M=992, K=640, N=640, residual=True, seed=13)                      # residual + a last panel with one block of three


# ---- base-weight gradients for full fine-tuning (csrc/full_grad.hip) ------------------------------------------------------------------
@pytest.mark.parametrize("mode,n_img,h,w,frames,c0,c1", [
    (1, 16, 40, 64, 0, 320, 0), (1, 2, 10, 16, 0, 640, 640), (2, 16, 40, 64, 0, 320, 0), (3, 16, 20, 32, 0, 640, 0), (4, 16, 10, 16, 16, 1280, 0),
    (4, 16, 40, 64, 16, 320, 0), (5, 2, 64, 64, 0, 128, 0), (1, 2, 8, 8, 0, 8, 0)])
def test_im2col_on_device(pair, mode, n_img, h, w, frames, c0, c1):
    """t2v_im2col_bf16 at the UNet's conv shapes (3x3, stride 2, nearest-x2, temporal, the VAE encoder's padded stride 2, the 8-channel
    entry conv) against the emulation (F.unfold): a copy — exact."""
    x0, x1 = pair.act(_rt(n_img * h * w, c0, seed=1)), (pair.act(_rt(n_img * h * w, c1, seed=2)) if c1 else (None, None))
    taps = 3 if mode == 4 else 9
    rows = pair.hip.im2col_rows(mode, n_img, h, w)
    assert rows == pair.emu.im2col_rows(mode, n_img, h, w)
    o_h = torch.full((rows, taps * (c0 + c1)), float("nan"), dtype=torch.bfloat16, device="cuda")
    o_e = torch.zeros(rows, taps * (c0 + c1))
    pair.run("im2col", (x0[0], x1[0], mode, n_img, h, w, frames, o_h), (x0[1], x1[1], mode, n_img, h, w, frames, o_e))
    assert torch.equal(o_h.float().cpu(), o_e)


@pytest.mark.parametrize("N,C,taps,kind", [(320, 320, 9, 0), (1280, 2560, 9, 0), (1280, 2560, 9, 1), (320, 320, 3, 1), (640, 1920, 3, 0),
                                           (4, 320, 9, 1), (70, 300, 9, 0), (130, 33, 3, 1)])
def test_repack_conv_on_device(pair, N, C, taps, kind):
    """t2v_repack_conv_f32 at the UNet's conv leaves (3x3 incl. the 2 560-channel concat inputs, (3,1,1), the 4-filter exit conv) and two
    ragged shapes: bit-identical to the torch permute / flip / cast chain that first makes the packs."""
    w = _rt(N, C * taps, seed=7).reshape(N, C, 3, 3) if taps == 9 else _rt(N, C * taps, seed=7).reshape(N, C, 3, 1, 1)
    shape = (N, taps * C) if kind == 0 else (C, taps * N)
    o_e = torch.zeros(shape)
    pair.emu.repack_conv(w, o_e, kind)
    o_h = torch.full(shape, float("nan"), dtype=torch.bfloat16, device="cuda")
    pair.hip.repack_conv(w.cuda(), o_h, kind)
    torch.cuda.synchronize()
    assert torch.equal(o_h.float().cpu(), o_e.bfloat16().float())


@pytest.mark.parametrize("kind,c0,c1,units,rows,silu,sum_rows", [
    (0, 320, 0, 16, 2560, True, 40960), (0, 640, 320, 1, 10240, True, 10240), (0, 1280, 0, 16, 160, False, 2560), (1, 320, 0, 1, 40960, False, 40960),
    (1, 1280, 0, 1, 2560, False, 2560), (2, 320, 0, 1, 40960, False, 40960), (2, 2560, 0, 1, 4096, False, 4096), (2, 640, 0, 1, 20480, False, 10240)])
def test_norm_affine_grad_on_device(pair, kind, c0, c1, units, rows, silu, sum_rows):
    """t2v_norm_affine_grad at the UNet's sizes: GroupNorm(+SiLU) with a two-part input, LayerNorm, bias column sums, per-clip sums."""
    M, Cc, G = units * rows, c0 + c1, 32
    x0, x1 = pair.act(_rt(M, c0, seed=1)), (pair.act(_rt(M, c1, seed=2)) if c1 else (None, None))
    dy = pair.act(_rt(M, Cc, seed=3))
    gamma, beta = pair.f32(_rt(Cc, seed=5) * 0.2 + 1.0), pair.f32(_rt(Cc, seed=6) * 0.1)
    n_out = M // sum_rows
    outs = []
    for side, ops in enumerate((pair.hip, pair.emu)):
        dev = "cuda" if side == 0 else "cpu"
        kw = dict(kind=kind, sum_rows=sum_rows, silu=silu)
        if kind == 0:
            stats = torch.zeros(units, 2 * G)
            pair.emu.gn_stats(x0[1], x1[1], units, rows, 1e-5, None, stats, G)
            kw.update(rows_per_unit=rows, groups=G, stats=stats.to(dev), gamma=gamma[side], beta=beta[side])
        elif kind == 1:
            kw.update(eps=1e-5)
        dg = torch.full((n_out, Cc), 3.0, device=dev) if kind != 2 else None
        db = torch.full((n_out, Cc), 3.0, device=dev)
        ws = torch.zeros(max(ops.norm_affine_grad_ws_floats(M, sum_rows, Cc), 1), device=dev)
        ops.norm_affine_grad(None if kind == 2 else x0[side], x1[side], dy[side], ws=ws, dgamma=dg, dbeta=db, **kw)
        outs.append((None if dg is None else dg.cpu(), db.cpu()))
    torch.cuda.synchronize()
    (dg_h, db_h), (dg_e, db_e) = outs
    assert rel_l2(db_h, db_e) < 2e-3      # (bf16 inputs on the device against fp32 copies of the same values: summation order only)
    if kind != 2:
        assert rel_l2(dg_h, dg_e) < 2e-3
