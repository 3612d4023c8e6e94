"""t2v_linear_pr (csrc/linear_pr.hip: activation panel resident in LDS, weights streamed into registers in MFMA fragment order) on the
host SIMT simulator, against the emulated backend (torch matmul on the unpacked weights): both panel geometries (K = 320 on 160-row
panels, K = 640 on 96-row panels), both epilogues (bias -> GEGLU; bias + optional residual), ragged row counts, fewer chunks than
waves, several chunks per wave, column splits over blockIdx.y, strided operands.

The simulator runs the kernel source itself; what it cannot show is the asynchrony of the weight ring (register loads in flight
across chunk boundaries) — tests/test_gpu_kernels.py::test_linear_pr_* cover that on the device."""
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
EMU = EmuOps()


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


def _case(sim, *, M, K, N, act=nt.ACT_NONE, bias=True, residual=False, ny=0, seed=0, lda=None, ldo=None, ln_in=False, gn_rpu=0):
    n_out = N // 2 if act == nt.ACT_GEGLU else N
    a_full = _rt(M, lda or K, seed=seed).bfloat16()
    if ln_in:   # rows with their own offsets and scales: a LayerNorm that mixed up rows or columns could not pass
        a_full = (a_full.float() * (0.5 + torch.arange(M)[:, None] % 7) + (torch.arange(M)[:, None] % 5 - 2.0)).bfloat16()
    a = a_full[:, :K]
    w = _rt(N, K, seed=seed + 1, scale=K ** -0.5).bfloat16()
    wp = nt.pack_linear_pr(w)
    assert torch.equal(nt.unpack_linear_pr(wp), w)
    b = _rt(N, seed=seed + 2) if bias else None
    res = _rt(M, n_out, seed=seed + 3).bfloat16() if residual else None
    kw = dict(M=M, N=N, bias=b, residual=res, act=act)
    if ln_in:
        kw["ln_in"] = (1.0 + 0.2 * _rt(K, seed=seed + 4), 0.3 * _rt(K, seed=seed + 5), 1e-5)
    if gn_rpu:   # a per-(unit, channel) affine as t2v_gn_coef_cs writes it: [units][2][K]
        units = M // gn_rpu
        kw["gn_in"] = (torch.stack([1.0 + 0.3 * _rt(units, K, seed=seed + 6), 0.5 * _rt(units, K, seed=seed + 7)], dim=1).contiguous(), gn_rpu)
    outs = []
    sim.lib.t2v_linear_pr_force_split(ny)
    try:
        for ops in (sim, EMU):
            out_full = torch.full((M, ldo or n_out), float("nan")).bfloat16()
            out = out_full[:, :n_out]
            assert ops.linear_pr_supported(a, wp, out, **kw) == 1, "the panel-resident kernel refuses the case"
            ops.linear_pr(a, wp, out, **kw)
            outs.append(out.float())
            if ldo:   # nothing outside the output columns is written
                assert torch.isnan(out_full[:, n_out:].float()).all()
    finally:
        sim.lib.t2v_linear_pr_force_split(0)
    y, r = outs
    assert torch.isfinite(y).all()
    assert rel_l2(y, r) < BF16_TOL, rel_l2(y, r)
    # (per-element: a permuted channel or a transposed block would pass a norm test on random data only by luck; this cannot)
    assert (y - r).abs().max() < 0.05 * r.abs().max()


def test_geglu_k320_several_chunks_per_wave(sim):
    # 20 chunks of 64 packed rows: waves 0-3 walk three chunks, 4-7 two (the weight ring runs across the chunk boundaries); M ragged
    _case(sim, M=160 + 75, K=320, N=1280, act=nt.ACT_GEGLU, seed=1)


def test_plain_k320_fewer_chunks_than_waves(sim):
    # N = 320: five chunks, three waves leave after the panel fill; residual and no residual; no bias
    _case(sim, M=320, K=320, N=320, residual=True, seed=2)
    _case(sim, M=160, K=320, N=320, bias=False, seed=3)


def test_plain_k320_qkv_width_and_strided_operands(sim):
    # N = 960 (15 chunks: wave 7 has one); A a column slice of a wider buffer, the output a column slice of a wider one
    _case(sim, M=200, K=320, N=960, seed=4, lda=384, ldo=1024)


def test_k640_on_96_row_panels(sim):
    _case(sim, M=96 * 2 + 64, K=640, N=640, residual=True, seed=5)   # (a residual needs M % 32 == 0: the last panel has two of its three blocks)
    _case(sim, M=96 * 2 + 40, K=640, N=640, seed=15)
    _case(sim, M=96, K=640, N=1024, act=nt.ACT_GEGLU, seed=6)


def test_k512_on_96_row_panels(sim):
    # K = 512 (32 steps: the four-slot weight ring): q | k | v and the GEGLU projection of the 8-head temporal transformer behind the entry
    # conv, with and without LayerNorm in the fill; a residual is not taken at this width
    _case(sim, M=96 * 2 + 40, K=512, N=1536, bias=False, seed=51)
    _case(sim, M=200, K=512, N=1024, act=nt.ACT_GEGLU, ln_in=True, seed=52)
    _case(sim, M=96, K=512, N=1536, bias=False, ln_in=True, ny=2, seed=53)
    a, out = _rt(96, 512).bfloat16(), torch.empty(96, 512).bfloat16()
    wp = nt.pack_linear_pr(_rt(512, 512).bfloat16())
    for ops in (sim, EMU):
        assert ops.linear_pr_supported(a, wp, out, M=96, N=512) == 1
        assert ops.linear_pr_supported(a, wp, out, M=96, N=512, residual=out) == 0


def test_column_splits_over_workgroup_rows(sim):
    # blockIdx.y: each workgroup row walks its own run of chunks with its own bias slice (also a ragged last run: 20 chunks in 3)
    _case(sim, M=170, K=320, N=1280, act=nt.ACT_GEGLU, ny=2, seed=7)
    _case(sim, M=128, K=640, N=1280, residual=True, ny=3, seed=8)


def test_layernorm_in_the_panel_fill(sim):
    # ln_in: norm1 / norm2 / norm3 -> one Linear (attention.py:300-311): both geometries, both epilogues, ragged rows (the rows past M are
    # normalised copies of the last row and written as zeros), uneven row groups per wave (20 row groups over 8 waves), column splits
    _case(sim, M=160 + 75, K=320, N=640, act=nt.ACT_GEGLU, ln_in=True, seed=31)
    _case(sim, M=200, K=320, N=960, ln_in=True, seed=32, lda=384, ldo=1024)
    _case(sim, M=96 + 40, K=640, N=640, ln_in=True, bias=False, seed=33)
    _case(sim, M=128, K=640, N=1280, act=nt.ACT_GEGLU, ln_in=True, ny=2, seed=34)


def test_groupnorm_affine_in_the_panel_fill(sim):
    # gn_in: x = proj_in(norm(x)) of the transformers (attention.py:373-389,471-513): several statistics units (each a whole number of
    # panels), both geometries, column splits; a unit that is not a whole number of panels is not taken
    _case(sim, M=640, K=320, N=320, gn_rpu=320, seed=41)
    _case(sim, M=320, K=320, N=640, gn_rpu=160, bias=False, seed=42, lda=384, ldo=704)
    _case(sim, M=384, K=640, N=640, gn_rpu=192, ny=2, seed=43)
    a = _rt(320, 320).bfloat16()
    out = torch.empty(320, 320).bfloat16()
    wp = nt.pack_linear_pr(_rt(320, 320).bfloat16())
    coef = torch.zeros(2, 2, 320)
    for ops in (sim, EMU):
        assert ops.linear_pr_supported(a, wp, out, M=320, N=320, gn_in=(coef, 160)) == 1
        assert ops.linear_pr_supported(a, wp, out, M=320, N=320, gn_in=(torch.zeros(4, 2, 320), 80)) == 0        # half a panel per unit
        assert ops.linear_pr_supported(a, wp, out, M=320, N=320, gn_in=(coef, 160), residual=out) == 0


def test_not_taken_cases(sim):
    a = _rt(64, 320).bfloat16()
    out = torch.empty(64, 128).bfloat16()
    wp = nt.pack_linear_pr(_rt(128, 320).bfloat16())
    assert sim.linear_pr_supported(a, wp, out, M=64, N=128) == 1
    for ops in (sim, EMU):
        assert ops.linear_pr_supported(a, wp, out, M=64, N=128, alpha=0.5) == 0
        assert ops.linear_pr_supported(a, wp, out, M=64, N=128, a1=a) == 0
        assert ops.linear_pr_supported(a, wp, out, M=64, N=128, rowvec=torch.zeros(1, 128), rowvec_div=64) == 0
        assert ops.linear_pr_supported(a[:, :256], wp, out, M=64, N=128) == 0          # K = 256
        assert ops.linear_pr_supported(a, wp, out, M=64, N=96) == 0                    # N % 64
        assert ops.linear_pr_supported(a, wp, out[:, :64], M=64, N=128, act=nt.ACT_GEGLU, residual=out[:, :64]) == 0
        assert ops.linear_pr_supported(a, wp, out, M=64, N=128, residual=out, ln_in=(torch.ones(320), torch.zeros(320), 1e-5)) == 0   # ln_in + residual
        assert ops.linear_pr_supported(a[:40], wp, out[:40], M=40, N=128, residual=out[:40]) == 0       # residual with M % 32 != 0
    with pytest.raises(nt.NativeError):
        sim.linear_pr(a, wp, out, M=64, N=128, alpha=0.5)
