"""Flash-style spatial-attention backward (csrc/attention_bwd.hip; hardware-validated since round 2) on the host SIMT simulator against the
emulated definition: tile tails in both sequences, several heads / images, V from the token-major buffer and from the per-head
[keys][64] layout."""
import os
import shutil
import sys

import pytest
import torch

from tests.emu_ops import EmuOps
from tests.util import rel_l2

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim"))


@pytest.fixture(scope="module")
def sim():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    import build as hostsim_build
    from tests.test_hostsim_kernels import HostSimOps
    ops = HostSimOps(hostsim_build.build_full())
    ops.tune, ops._ws = {}, {}
    ops.init()
    return ops


def _rt(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).bfloat16().float()


def _tposed(t, n_img, seq, cols):
    sp = (seq + 63) // 64 * 64
    out = torch.zeros(n_img * cols, sp)
    for i in range(n_img):
        out[i * cols:(i + 1) * cols, :seq] = t[i * seq:(i + 1) * seq].t()
    return out


@pytest.mark.parametrize("n_img,seq,heads,v_layout", [(1, 64, 1, "tok"), (2, 100, 2, "tok"), (1, 200, 1, "head"), (1, 130, 3, "tok"),
                                                      (2, 40, 4, "head"), (1, 160, 5, "tok"), (1, 7, 1, "tok")])  # 40 / 160: the 5x8 and 10x16 levels
def test_attn_spatial_bwd(sim, n_img, seq, heads, v_layout):
    emu = EmuOps()
    inner, M = heads * 64, n_img * seq
    q, k, v = _rt(M, inner, seed=1, scale=0.8), _rt(M, inner, seed=2, scale=0.8), _rt(M, inner, seed=3)
    do = _rt(M, inner, seed=4)
    scale = 0.125
    o = torch.zeros(M, inner)
    for img in range(n_img):
        r = slice(img * seq, (img + 1) * seq)
        for hd in range(heads):
            c = slice(hd * 64, (hd + 1) * 64)
            o[r, c] = (q[r, c] @ k[r, c].t() * scale).softmax(dim=1) @ v[r, c]
    o = o.bfloat16().float()
    sp = (seq + 63) // 64 * 64
    if v_layout == "tok":
        v_buf, vis, vhs = v, seq * inner, 64
    else:  # per (image, head): [padded keys][64], the transpose of the forward's V^T rows
        v_buf = torch.zeros(n_img * heads * sp, 64)
        for img in range(n_img):
            for hd in range(heads):
                v_buf[(img * heads + hd) * sp:(img * heads + hd) * sp + seq] = v[img * seq:(img + 1) * seq, hd * 64:(hd + 1) * 64]
        vis, vhs = heads * sp * 64, sp * 64
    g_e = [torch.zeros(M, inner) for _ in range(3)]
    emu.attn_spatial_bwd(q, k, v_buf, vis, vhs, None, None, None, do, o, None, None, *g_e, n_img, seq, heads, scale)
    bf = lambda t: t.bfloat16().contiguous()  # noqa: E731
    g_s = [torch.full((M, inner), float("nan"), dtype=torch.bfloat16) for _ in range(3)]
    l2, ds = torch.zeros(n_img * heads, sp), torch.zeros(n_img * heads, sp)
    sim.attn_spatial_bwd(bf(q), bf(k), bf(v_buf), vis, vhs, bf(_tposed(k, n_img, seq, inner)), bf(_tposed(q, n_img, seq, inner)),
                         bf(_tposed(do, n_img, seq, inner)), bf(do), bf(o), l2, ds, *g_s, n_img, seq, heads, scale)
    for name, a, b in zip(("dq", "dk", "dv"), g_s, g_e):
        assert torch.isfinite(a.float()).all(), name
        assert rel_l2(a.float(), b) < 1.2e-2, (name, rel_l2(a.float(), b))


def test_results_do_not_depend_on_the_fiber_schedule(sim, monkeypatch):
    """The simulator visits runnable fibers round-robin by default; HOSTSIM_ORDER = reverse / random changes the interleaving of
    waves.  A kernel whose waves are correctly synchronised gives bit-identical results under every schedule — a missing barrier
    (one wave restaging an LDS tile another still reads) would not.  Checked for the two MFMA kernels written against the
    simulator, the GEMM and the LDS-transposing kernels."""
    from t2v_turbo_amd import native as nt
    n_img, seq, heads = 1, 130, 2
    inner, M = heads * 64, n_img * seq
    q, k, v, do = (_rt(M, inner, seed=s_, scale=0.8) for s_ in (1, 2, 3, 4))
    o = torch.zeros(M, inner)
    for hd in range(heads):
        c = slice(hd * 64, (hd + 1) * 64)
        o[:, c] = (q[:, c] @ k[:, c].t() * 0.125).softmax(dim=1) @ v[:, c]
    bf = lambda t: t.bfloat16().contiguous()  # noqa: E731
    a, b = _rt(300, 192, seed=5, scale=0.3), _rt(300, 64, seed=6, scale=0.3)
    ga, gw = _rt(200, 128, seed=7), _rt(96, 9 * 128, seed=8, scale=0.05)

    def run():
        g = [torch.zeros(M, inner, dtype=torch.bfloat16) for _ in range(3)]
        sp = 192
        sim.attn_spatial_bwd(bf(q), bf(k), bf(v), seq * inner, 64, bf(_tposed(k, 1, seq, inner)), bf(_tposed(q, 1, seq, inner)),
                             bf(_tposed(do, 1, seq, inner)), bf(do), bf(o.bfloat16().float()), torch.zeros(heads, sp), torch.zeros(heads, sp),
                             *g, n_img, seq, heads, 0.125)
        w_out = torch.zeros(192, 64)
        sim.wgrad_tn(bf(a), bf(b), w_out, alpha=0.5, splits=3)
        t_out = torch.zeros(192, 320, dtype=torch.bfloat16)
        sim.transpose_pad(bf(a), 300, 192, t_out)
        c_out = torch.zeros(2 * 10 * 10, 96, dtype=torch.bfloat16)
        sim.gemm(bf(ga), bf(gw), c_out, M=200, N=96, mode=nt.GEMM_CONV3X3, n_img=2, h=10, wd=10, tile_cfg=27)
        return [t.clone() for t in g] + [w_out, t_out, c_out]

    base = run()
    for order in ("reverse", "random"):
        monkeypatch.setenv("HOSTSIM_ORDER", order)
        for x, y in zip(base, run()):
            assert torch.equal(x, y), order
