"""t2v_conv_halo (csrc/conv_halo.hip: LDS-resident halo slab, 16x16x32 MFMA, k-groups) on the host SIMT simulator, against the
emulated backend (torch conv on the unpacked weights): every workgroup tile (320x160 one k-group, 320x80 two, 160x80 four on 16- and
32-wide grids), virtual concat, ragged rows / channels, padded weight stages, the epilogue terms and the column statistics.

The simulator runs the kernel source itself; what it cannot show is the asynchrony of the DMA ring (the GPU suite covers that:
tests/test_gpu_kernels.py::test_conv_halo_*)."""
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


def _case(sim, *, n_img, h, w, c0, N, c1=0, cfg=0, bias=True, rowvec=False, residual=False, act=0, colstat=False, seed=0, ups=0):
    """``ups`` = 1: T2V_GEMM_CONV3X3_UP2 — (h, w) is the SOURCE grid, the output grid is (2h, 2w)."""
    mode = nt.GEMM_CONV3X3_UP2 if ups else nt.GEMM_CONV3X3
    m_src = n_img * h * w
    M = m_src << (2 * ups)
    K = 9 * (c0 + c1)
    a0 = _rt(m_src, c0, seed=seed).bfloat16()
    a1 = _rt(m_src, c1, seed=seed + 1).bfloat16() if c1 else None
    wt = _rt(N, K, seed=seed + 2, scale=K ** -0.5).bfloat16()
    ws = nt.pack_conv_slab(wt)
    assert ws.shape[1] == nt.conv_halo_pack_cols(c0 + c1) == sim.lib.t2v_conv_halo_pack_cols(c0 + c1)
    assert torch.equal(nt.unpack_conv_slab(ws, c0 + c1), wt)
    b = _rt(N, seed=seed + 3) if bias else None
    div = (h * w) << (2 * ups)
    rv = _rt(M // div, N, seed=seed + 4) if rowvec else None
    res = _rt(M, N, seed=seed + 5).bfloat16() if residual else None
    kw = dict(M=M, N=N, a1=a1, mode=mode, n_img=n_img, h=h, wd=w, bias=b, rowvec=rv, rowvec_div=div if rowvec else 0,
              residual=res, act=act, tile_cfg=cfg)
    outs, stats = [], []
    for ops in (sim, EMU):
        out = torch.full((M, N), float("nan")).bfloat16()
        cs = torch.full((M // 32, N, 2), float("nan")) if colstat else None
        kk = dict(kw, colstat=cs) if colstat else dict(kw)
        assert ops.conv_halo_supported(a0, ws, out, **kk) == 1, "halo kernel refuses the case"
        ops.conv_halo(a0, ws, out, **kk)
        outs.append(out.float())
        stats.append(cs)
    y, r = outs
    assert torch.isfinite(y).all()
    assert rel_l2(y, r) < BF16_TOL, rel_l2(y, r)
    if colstat:   # statistics are of the bf16 values the kernel itself stored
        yo = y.reshape(M // 32, 32, N)
        want = torch.stack([yo.sum(dim=1), (yo * yo).sum(dim=1)], dim=2)
        assert torch.allclose(stats[0], want, rtol=1e-4, atol=1e-3), (stats[0] - want).abs().max()


# (tile id 40 = 320x160 / one k-group, 41 = 320x80 / two, 43 = 160x80 / four, all on 32-wide tiles; 42 = 160x80 on a 16-wide grid)
@pytest.mark.parametrize("cfg", [40, 41, 43])
def test_conv3x3_every_tile(sim, cfg):
    # 20x32 grid: two (cfg 40/41: 10 rows) or four (cfg 43: 5 rows) tiles per image; N = 160: 1 / 2 / 2 channel tiles
    _case(sim, n_img=2, h=20, w=32, c0=64, N=160, cfg=cfg, rowvec=True, residual=True, colstat=True, seed=cfg)


def test_conv3x3_two_tile_columns(sim):
    # 64-wide grid: two 32-column tiles per row of tiles (the halo crosses the tile boundary inside the image)
    _case(sim, n_img=1, h=10, w=64, c0=64, N=80, cfg=41, residual=True, colstat=True, seed=8)


def test_conv3x3_virtual_concat_and_padded_stages(sim):
    # c0 + c1 = 192: 6 sub-slabs x 9 taps = 54 pairs (not a multiple of the 4-pair stage: the pack's zero padding is multiplied)
    _case(sim, n_img=1, h=10, w=32, c0=128, c1=64, N=80, cfg=41, seed=3)
    _case(sim, n_img=1, h=10, w=32, c0=64, c1=128, N=80, cfg=43, seed=9)


def test_conv3x3_16_wide_grid_whole_frame_tiles(sim):
    # the 10x16 level: one 160-token tile per frame, four k-groups
    _case(sim, n_img=3, h=10, w=16, c0=64, N=160, cfg=42, rowvec=True, colstat=True, act=nt.ACT_SILU, seed=4)


def test_conv3x3_ragged_rows_and_channels(sim):
    # 12 rows with 10-row tiles (the second tile is mostly outside the image), N = 48 < 80 (ragged channel tile)
    _case(sim, n_img=1, h=12, w=32, c0=64, N=48, cfg=41, residual=True, seed=5)


def test_conv3x3_vae_decoder_widths(sim):
    # the KL-VAE decoder's widths (ae_modules.py:183-203): 128 / 256 / 512 channels are not multiples of the 80-channel wave tile;
    # the last channel tile is padded (128 = 80 + 48 or one 160-wide tile, 256 = 160 + 96, 512 = 6 x 80 + 32), with the epilogue
    # variants the decoder uses (residual; column statistics of the next GroupNorm)
    _case(sim, n_img=1, h=10, w=32, c0=128, N=128, residual=True, colstat=True, seed=21)
    _case(sim, n_img=1, h=20, w=32, c0=64, N=256, colstat=True, seed=22)
    _case(sim, n_img=1, h=10, w=32, c0=64, N=512, cfg=41, residual=True, colstat=True, seed=23)


def test_conv3x3_64_channel_wave_tiles(sim):
    # tile id 44: 320x128 on 80 x 64 wave tiles (four channel blocks per wave) — what the heuristic picks for N = 128 / 256 / 512;
    # forced here on a ragged width too (N = 192: the second channel tile is half empty), with a row vector and SiLU
    _case(sim, n_img=1, h=10, w=32, c0=128, N=128, cfg=44, residual=True, colstat=True, seed=31)
    _case(sim, n_img=1, h=20, w=32, c0=64, c1=64, N=256, cfg=44, rowvec=True, colstat=True, seed=32)
    _case(sim, n_img=1, h=12, w=64, c0=64, N=192, cfg=44, rowvec=True, act=nt.ACT_SILU, seed=33)


def test_conv3x3_over_nearest_x2_upsampled_source(sim):
    # T2V_GEMM_CONV3X3_UP2 (Upsample: interpolate x2, then conv): the slab is gathered from the half-size source grid
    # 10x16 -> 20x32 (two 10-row tiles per image), with the epilogue variants; 5x16 -> 10x32 on the 64-channel wave tiles; a source
    # whose output grid is ragged against the tile rows (6x16 -> 12x32)
    _case(sim, n_img=2, h=10, w=16, c0=64, N=160, cfg=40, rowvec=True, colstat=True, seed=41, ups=1)
    _case(sim, n_img=1, h=5, w=16, c0=128, N=128, cfg=44, residual=True, colstat=True, seed=42, ups=1)
    _case(sim, n_img=1, h=6, w=16, c0=64, c1=64, N=80, cfg=41, act=nt.ACT_SILU, seed=43, ups=1)
    _case(sim, n_img=3, h=5, w=8, c0=64, N=160, cfg=42, colstat=True, seed=44, ups=1)      # 10x16 output: whole-frame tiles, four k-groups


def test_heuristic_picks_a_tile_and_matches(sim):
    _case(sim, n_img=2, h=10, w=16, c0=64, N=80, seed=6)
    _case(sim, n_img=2, h=20, w=32, c0=128, N=160, seed=7)


def test_not_taken_cases(sim):
    a0 = _rt(64, 64).bfloat16()
    w = nt.pack_conv_slab(_rt(32, 64 * 9).bfloat16())
    out = torch.empty(64, 32).bfloat16()
    # width not a multiple of 16, the temporal conv, linear mode
    assert sim.conv_halo_supported(a0, w, out, M=64, N=32, mode=nt.GEMM_CONV3X3, n_img=1, h=8, wd=8) == 0
    assert EMU.conv_halo_supported(a0, w, out, M=64, N=32, mode=nt.GEMM_CONV3X3, n_img=1, h=8, wd=8) == 0
    assert sim.conv_halo_supported(a0, w, out, M=64, N=32, mode=nt.GEMM_LINEAR) == 0
    assert sim.conv_halo_supported(a0, w, out, M=64, N=32, mode=nt.GEMM_TCONV3, n_img=16, h=2, wd=2, frames=16) == 0
