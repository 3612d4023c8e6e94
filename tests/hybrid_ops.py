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
             "softmax_bwd_rows", "transpose", "transpose_pad", "sumpool2x2", "gather", "dropout")


class HybridOps(EmuOps):
    def __init__(self):
        super().__init__(act_dtype=torch.bfloat16, strict=True)
        import build as hostsim_build
        from tests.test_hostsim_kernels import HostSimOps
        self.sim = HostSimOps(hostsim_build.build())
        self.sim_calls = 0
        for name in SIMULATED:
            setattr(self, name, self._route(name))

    def _route(self, name):
        fn = getattr(self.sim, name)

        def call(*a, **k):
            self.sim_calls += 1
            self._log(name)
            return fn(*a, **k)
        return call
