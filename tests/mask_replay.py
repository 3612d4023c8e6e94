"""TEST ONLY: replay the native gradient engine's counter-based dropout masks inside the torch module.

The train-mode student (train_t2v_turbo_v1_lora.py:641) has Dropout(0.1) on every LoRA branch (utils/lora.py:45-50,124-129)
and in the temporal conv blocks (openaimodel3d.py:257-309).  The engine draws its masks from (step seed, site, element)
(csrc/train.hip), torch from its own random stream: parity is checked by patching every ``nn.Dropout`` the engine applied
with the engine's mask, re-laid into the row / channel order torch sees at that point.  Shared by the CPU suite (emulated
backend, masks captured as they are drawn) and the GPU suite (masks regenerated on the host from the recorded site
geometry: ``tests.emu_ops.EmuOps.dropout_keep`` is bit-identical to the device mask, tests/test_gpu_unet_grad.py)."""
import torch
from t2v_turbo_amd import native as nt


def patch_engine_masks(m, eng, masks):
    """``masks``: {site id: bool keep [rows, ncols]} for every entry of ``eng.drop_sites``.  Patches the Dropout modules of
    ``m`` (a module with the same structure as ``eng.model``; modules are matched by traversal order)."""
    from t2v_turbo_amd.unet3d import GEGLU
    src_mods = list(eng.model.modules())
    dst_mods = list(m.modules())
    assert len(src_mods) == len(dst_mods)
    twin = {id(a): b for a, b in zip(src_mods, dst_mods)}
    leaf_of = {id(mod.dropout): mod for mod in m.modules() if hasattr(mod, "lora_up")}
    geglu = {id(mod.proj) for mod in m.modules() if isinstance(mod, GEGLU)}

    def patch(drop, keep, kind, meta, p):
        def fwd(t):
            if kind == "rows":
                k = keep.reshape(t.shape)
            elif kind == "temporal":
                B, F, hw = meta
                k = keep.view(B, F, hw, -1).permute(0, 2, 1, 3).reshape(t.shape)
            elif kind == "ctx":
                B, F, L = meta
                k = keep.view(B, 1, L, -1).expand(B, F, L, keep.shape[1]).reshape(t.shape)
            elif kind == "conv":
                n, ho, wo = meta
                k = keep.view(n, ho, wo, -1).permute(0, 3, 1, 2)
            else:
                B, F, h, w = meta
                k = keep.view(B, F, h, w, -1).permute(0, 4, 1, 2, 3)
            assert k.shape == t.shape, (kind, k.shape, t.shape)
            return t * k * nt.dropout_inv_keep(p)   # the engine's scale: 1 / (1 - thr16 / 65536)
        drop.forward = fwd

    for sid, (drops, kind, meta) in enumerate(eng.drop_sites):
        keep = masks[sid]
        c0 = 0
        for d_src in drops:
            d = twin[id(d_src)]
            if kind == "tconv":
                patch(d, keep, kind, meta, d.p)
                continue
            leaf = leaf_of[id(d)]
            n_out = leaf.lora_up.weight.shape[0]
            k = keep[:, c0:c0 + n_out]
            if id(leaf) in geglu:  # the engine's columns are the packed GEGLU rows
                j = torch.arange(n_out)
                perm = (j // 64) * 32 + (j % 32) + (j % 64 >= 32) * (n_out // 2)
                full = torch.empty_like(k)
                full[:, perm] = k
                k = full
            patch(d, k.contiguous(), kind, meta, d.p)
            c0 += n_out
        assert kind == "tconv" or c0 == keep.shape[1]


class SiteGeometrySpy:
    """Wraps a HipOps instance's ``dropout`` / ``gemm`` so that the (rows, ncols, p) of every dropout site is noted while the
    engine records its launch lists (the device keeps no masks: they are a function of seed, site and element index)."""

    def __init__(self, ops):
        self.sites = {}
        o_drop, o_gemm = ops.dropout, ops.gemm

        def dropout(x, resid, out, ncols, p, seed, site):
            self._note(site, x.shape[0], ncols, p)
            return o_drop(x, resid, out, ncols, p, seed, site)

        def gemm(a0, w, out, **kw):
            dr = kw.get("dropout")
            if dr is not None and dr[0] > 0:
                self._note(dr[2], kw["M"], dr[3], dr[0])
            return o_gemm(a0, w, out, **kw)

        ops.dropout, ops.gemm = dropout, gemm

    def _note(self, site, rows, ncols, p):
        prev = self.sites.setdefault(int(site), (int(rows), int(ncols), float(p)))
        assert prev == (int(rows), int(ncols), float(p)), (site, prev, rows, ncols, p)

    def masks(self, seed):
        from tests.emu_ops import EmuOps
        return {s: EmuOps.dropout_keep(int(seed), s, r, c, p) for s, (r, c, p) in self.sites.items()}
