"""NOT collected by default (the file name does not match test_*.py): the GPU test of t2v_gemm's LoRA epilogue (lora_* fields,
csrc/gemm_fuse.hip FUSE bit 16), written after round 3's GPU budget was spent.  First thing to run next round:

    python -m pytest tests/gpu_unvalidated_lora_epilogue.py -m gpu -q

The same cases pass on the host SIMT simulator (tests/test_hostsim_gemm_fuse.py::test_lora_branch_in_the_base_leaf_epilogue)."""
import pytest
import torch

from t2v_turbo_amd import native as nt
from tests.emu_ops import EmuOps
from tests.util import rel_l2

pytestmark = pytest.mark.gpu
EMU = EmuOps()
BF16_TOL = 4e-3


def _rt(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).bfloat16().float()


def _dev(t):
    return None if t is None else t.bfloat16().cuda().contiguous()


@pytest.mark.parametrize("M,K,C,leaves", [(40960, 320, 320, 3), (40960, 320, 320, 1), (10240, 640, 640, 1), (2560, 1280, 1280, 3), (1000, 128, 96, 3)])
def test_lora_branch_in_the_base_leaf_epilogue(M, K, C, leaves):
    from t2v_turbo_amd.native import HipOps
    hip = HipOps()
    hip.init()
    N, p, site = leaves * C, 0.1, 4
    seed = torch.tensor([0x5EED_1234_ABCD], dtype=torch.int64)
    x, w, b = _rt(M, K, seed=1), _rt(N, K, seed=2, scale=K ** -0.5), _rt(N, seed=3)
    res, t, u = _rt(M, N, seed=4), _rt(M, leaves * 64, seed=5, scale=0.5), _rt(N, 64, seed=6, scale=0.2)
    for drop in (None, (p, seed, site, N, 0)):
        o_h, o_e = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda"), torch.zeros(M, N)
        kw = dict(M=M, N=N, bias=b)
        drop_h = None if drop is None else (p, seed.cuda(), site, N, 0)
        lo_h = (_dev(t), _dev(u), C, 0.5)
        if not hip.gemm_fuse_supported(_dev(x), _dev(w), o_h, residual=_dev(res), lora=lo_h, bias=b.cuda(), M=M, N=N, dropout=drop_h):
            pytest.skip("this shape's tuned tile has no fused twin")
        hip.gemm(_dev(x), _dev(w), o_h, residual=_dev(res), lora=lo_h, bias=b.cuda(), M=M, N=N, dropout=drop_h)
        EMU.gemm(x, w, o_e, residual=res, lora=(t, u, C, 0.5), dropout=drop, **kw)
        torch.cuda.synchronize()
        got = o_h.float().cpu()
        assert torch.isfinite(got).all() and rel_l2(got, o_e) < BF16_TOL, (drop is not None, rel_l2(got, o_e))
        o2 = torch.zeros_like(o_h)
        hip.gemm(_dev(x), _dev(w), o2, residual=_dev(res), lora=lo_h, bias=b.cuda(), M=M, N=N, dropout=drop_h)
        torch.cuda.synchronize()
        assert torch.equal(o2, o_h)
