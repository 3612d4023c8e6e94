"""TEST ONLY: the engine's op backend with every SIMT-only kernel running as REAL kernel source on the host SIMT simulator
(tests/hostsim) in bf16, and the rest (GEMM family, forward attention / norms: MFMA, asm or validated long ago) on the torch
emulation rounding to bf16.  An engine recorded against it exercises the ctypes marshalling of ``native.HipOps`` (row strides
of column slices, alignment rules the C entry points enforce) and the bf16 numerics of the whole path — what the first GPU
run would otherwise be the first to see."""
import os
import sys

import torch

from tests.emu_ops import EmuOps

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim"))

SIMULATED = ("gn_bwd", "gn_bwd_ws_floats", "layernorm_bwd", "geglu_fwd", "geglu_bwd", "scatter2x", "add", "attn_temporal_bwd",
             "softmax_bwd_rows", "transpose", "transpose_pad", "sumpool2x2", "gather", "dropout", "wgrad_tn", "wgrad_tn_group")


class HybridOps(EmuOps):
    """``real_gemm=True`` additionally runs every t2v_gemm launch of the engine as the real MFMA kernel source on the simulator
    (minutes instead of seconds: opt-in)."""

    def __init__(self, real_gemm=False):
        super().__init__(act_dtype=torch.bfloat16, strict=True)
        import build as hostsim_build
        from tests.test_hostsim_kernels import HostSimOps
        self.sim = HostSimOps(hostsim_build.build())
        self.sim_calls = 0
        for name in SIMULATED:
            setattr(self, name, self._route(name))
        if real_gemm:
            self.gsim = HostSimOps(hostsim_build.build_gemm())
            self.gsim.tune, self.gsim._ws = {}, {}
            self.gemm_calls = 0
            emu_gemm = self.gemm

            def gemm(a0, w, out, **kw):
                self.gemm_calls += 1
                self._log("gemm")
                if out.dtype not in (torch.bfloat16, torch.float32) or a0.dtype != torch.bfloat16:
                    return emu_gemm(a0, w, out, **kw)
                b, rv = kw.get("bias"), kw.get("rowvec")
                kw["bias"] = None if b is None else b.float().contiguous()
                if rv is not None and rv.dtype != torch.float32:
                    kw["rowvec"] = rv.float()
                return self.gsim.gemm(a0, w, out, **kw)
            self.gemm = gemm

    def _route(self, name):
        fn = getattr(self.sim, name)

        def call(*a, **k):
            self.sim_calls += 1
            self._log(name)
            return fn(*a, **k)
        return call
