"""Base-weight gradients of the student UNet on the native gradient engine: FULL fine-tuning, the call pattern of
``train_latent_t2v_turbo_v2.py`` (``:669`` ``unet.requires_grad_(True).train()``, ``:798-816`` every UNet parameter in an optimizer
group, ``:1262`` ``accelerator.backward``) — no LoRA injection, every leaf trainable.

The tape and the data gradients are the gradient engine's (engine_unet_bwd.py).  This mixin adds, at every leaf's backward,

  Linear / 1x1 conv        dW = dy^T x                     ``t2v_wgrad_tn`` on the token-major operands (per leaf of a q | k | v group,
                                                            per input part of a virtual concat); GEGLU: rows un-permuted by one gather
  3x3 / strided / x2 /     dW[n][tap][c] = dy^T xcol       ``t2v_im2col_bf16`` (the shifted-row matrix in the forward pack's K order) +
  (3,1,1) conv                                              ``t2v_wgrad_tn`` + one ``t2v_gather_f32`` into the parameter's [N, C, k..] layout
  bias                     db = column sums of dy          ``t2v_norm_affine_grad`` kind 2
  GroupNorm / LayerNorm    dgamma, dbeta                   ``t2v_norm_affine_grad`` kinds 0 / 1 (SiLU behind the norm included)
  time-embedding rows      d(loss)/d(emb_all)              per-clip column sums of the ResBlock's conv gradient (kind 2, sum_rows = F h w)

into one fp32 tensor per parameter (``plan["fgrads"]``), which the autograd bridge (unet3d._NativeStudentFull) hands to torch.
The conditioning branch (time / fps / guidance MLPs, ``emb_layers``: B rows) stays with torch autograd behind ``emb_all``, as in LoRA
training.  The weights change every optimizer step: the Packer re-fills its packs IN PLACE (``Packer.refresh``), so the recorded launch
lists — raw device pointers — stay valid and nothing is re-recorded.

Correctness first: im2col is materialised (2 taps C bytes per output row) and every leaf's gradient is its own launches; the step is
not a measured configuration of bench.py.  Verified on CPU against torch autograd through the module (tests/test_unet_full_grad_cpu.py),
on the host simulator per kernel, and on MI355X against the imported reference's own parameter gradients (tests/golden/unet_tiny_full_grad.npz)."""
import os

import torch
import torch.nn as nn

from . import native as nt
from .engine import is_lora_leaf


class FullTrainMixin:
    full_params = None

    def bind_full(self, params):
        """``params``: the parameters whose gradients the engine computes (every UNet parameter outside the conditioning branch)."""
        self.full_params = list(params)
        self.full_ids = {id(p) for p in self.full_params}
        self.plans.clear()
        self.fingerprint = None
        self._full_fp = None

    @property
    def training_full(self):
        return self.full_params is not None

    @property
    def trains(self):
        """LoRA training or full fine-tuning: the forward keeps what weight gradients need, dropouts are applied, emb_all is an input."""
        return self.training_full or getattr(self, "lora_params", None) is not None

    @staticmethod
    def conditioning_module_ids(model):
        """Modules of the B-row conditioning branch (their parameters' gradients are torch's, through ``emb_all``)."""
        from .unet3d import ResBlock
        cond = set()
        for name in ("time_embed", "fps_embedding", "time_cond_proj", "motion_cond_proj", "combine_proj"):
            sub = getattr(model, name, None)
            if sub is not None:
                cond.update(id(x) for x in sub.modules())
        for mod in model.modules():
            if isinstance(mod, ResBlock):
                cond.update(id(x) for x in mod.emb_layers.modules())
        return cond

    @classmethod
    def engine_parameters(cls, model):
        """Every parameter of ``model`` the engine differentiates, in ``named_parameters`` order."""
        cond = cls.conditioning_module_ids(model)
        out, seen = [], set()
        for mod in model.modules():
            if id(mod) in cond:
                continue
            for p in mod._parameters.values():
                if p is not None and id(p) not in seen:
                    seen.add(id(p))
                    out.append(p)
        return out

    # ---- per-step pack refresh ------------------------------------------------------------------------------------------
    def full_refresh_packs(self):
        """Re-fill the weight packs in place when a parameter changed since the last forward (the launch lists keep their pointers)."""
        from .engine import params_fingerprint
        fp = params_fingerprint(self.model)
        if getattr(self, "_full_fp", None) is None:
            self._full_fp = fp
        elif fp != self._full_fp:
            self._refresh()
            self._full_fp = fp

    # The refresh is ~ 2 000 small launches (torch copies / casts, library transposes and repacks) over FIXED tensors — the parameters'
    # storages in, the packs out — so after one eager pass it is captured as ONE hipGraph and replayed per optimizer step: the host's
    # 15-17 ms of issuing it (tools/r6_gpu_calls/README.md, call 40) go.  The graph holds raw pointers: it is dropped and re-captured
    # when a parameter was re-homed or a pack was added; T2V_REFRESH_GRAPH=0 keeps the eager loop.
    refresh_graph = os.environ.get("T2V_REFRESH_GRAPH", "1") == "1"

    def _refresh(self):
        dev = getattr(self, "device", None)
        if not (self.refresh_graph and getattr(self.ops, "is_native", False) and dev is not None and dev.type == "cuda"):
            self.pk.refresh(self.ops)
            return
        from .nn_util import walk_parameters
        sig = (len(self.pk.makers), tuple(p.data_ptr() for p in walk_parameters(self.model)))
        st = getattr(self, "_refresh_state", None)
        if st is not None and st["sig"] == sig and st["graph"] is not None:
            st["graph"].replay()
            return
        if st is None or st["sig"] != sig:
            self.pk.refresh(self.ops)               # first pass for this set of tensors: eager (it also warms the allocator)
            self._refresh_state = {"sig": sig, "graph": None, "failed": False}
            return
        if st["failed"]:
            self.pk.refresh(self.ops)
            return
        try:                                         # second pass: capture, then run the captured pass
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize(dev)
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.pk.refresh(self.ops)
            g.replay()
            st["graph"] = g
        except Exception as e:  # noqa: BLE001 - a maker the capture cannot take (host read-back, pageable copy): stay eager, say so once
            import warnings
            warnings.warn(f"pack refresh not capturable as a hipGraph ({type(e).__name__}: {e}); staying with the eager pass")
            st["failed"] = True
            torch.cuda.synchronize(dev)
            self.pk.refresh(self.ops)

    # ---- forward side: keep the leaf's input ----------------------------------------------------------------------------
    def full_save(self, mods, x, **info):
        if any(is_lora_leaf(mm) for mm in mods):
            raise NotImplementedError("native full fine-tuning of a LoRA-injected network (train the LoRA tensors, or merge them first)")
        key = tuple(id(mm) for mm in mods)
        old = self._fsaved.pop(key, None)
        if old is not None:   # (a checkpointed block's recomputation saves again)
            self.drop(*old[0].parts)
        self.hold(*x.parts)
        self._fsaved[key] = (x, info)

    def full_take(self, mods):
        x, info = self._fsaved.pop(tuple(id(mm) for mm in mods))
        return x, info

    def fgrad(self, p):
        """The fp32 gradient tensor of parameter ``p``: a view into the plan's gradient arena (one allocation for all parameters, made at the
        first request: recorded launches point into it, and the hand-over to torch is ONE copy of the arena instead of one per tensor)."""
        g = self.plan["fgrads"].get(id(p))
        if g is None:
            assert id(p) in self.full_ids, "a leaf parameter that was not given to bind_full"
            if self.plan.get("fgrad_arena") is None:
                off, offs = 0, {}
                for q in self.full_params:
                    offs[id(q)] = off
                    off += (q.numel() + 63) // 64 * 64        # (256-byte aligned views)
                self.plan["fgrad_off"] = offs
                self.plan["fgrad_arena"] = torch.zeros(off, dtype=torch.float32, device=self.device)
            o = self.plan["fgrad_off"][id(p)]
            g = self.plan["fgrads"][id(p)] = self.plan["fgrad_arena"][o:o + p.numel()].view(p.shape)
        return g

    def _idx(self, key, make):
        return self.pk._memo(("full_idx",) + key, lambda: make().to(torch.int32).to(self.device).contiguous())

    def full_wgrad(self, dy, xmat, out):
        """out[R, C] = dy^T xmat by t2v_wgrad_tn (one launch pair, or one launch where the product has enough output tiles to fill the
        chip without splitting the token range — then it needs no partial slabs, however large the output: 1 280 x 23 040 fp32 = 118 MB
        for the 2 560-channel 3x3 convs of the decoder half)."""
        self.ops.wgrad_tn(dy, xmat, out)

    def full_colsum(self, dy, dst, sum_rows):
        """dst[u][c] = sum of ``sum_rows`` consecutive rows of dy (fp32 [rows / sum_rows, C], a column slice of a wider buffer allowed)."""
        ops = self.ops
        ws = self.buf(1, max(ops.norm_affine_grad_ws_floats(dy.shape[0], sum_rows, dy.shape[1]), 1), torch.float32)
        ops.norm_affine_grad(None, None, dy, kind=2, sum_rows=sum_rows, ws=ws, dbeta=dst)
        self.pool.put(ws)

    def full_bias_grad(self, bias, dy, inv_perm=None):
        if bias is None:
            return
        g = self.fgrad(bias)
        if inv_perm is None and dy.shape[1] == bias.numel():
            self.full_colsum(dy, g.view(1, -1), dy.shape[0])
            return
        tmp = self.buf(1, dy.shape[1], torch.float32)
        self.full_colsum(dy, tmp, dy.shape[0])
        idx = inv_perm if inv_perm is not None else self._idx(("head", bias.numel()), lambda: torch.arange(bias.numel()))
        self.ops.gather(tmp, idx, g)
        self.pool.put(tmp)

    def full_linear_grads(self, mods, dy):
        """dW (and db) of the Linear / 1x1-conv leaves ``mods`` whose row-concatenated weights produced dy's columns."""
        ops = self.ops
        x, info = self.full_take(mods)
        perm = info.get("perm")
        off = 0
        for mod in mods:
            w, b = mod.weight, mod.bias
            n, k = w.shape[0], w.numel() // w.shape[0]
            d = dy[:, off:off + n]
            off += n
            if perm is None:
                g2 = self.fgrad(w).view(n, k)
                c0 = x.parts[0].shape[1]
                self.full_wgrad(d, x.parts[0], g2[:, :c0])
                if x.p1 is not None:
                    self.full_wgrad(d, x.p1, g2[:, c0:])
                self.full_bias_grad(b, d)
            else:
                # packed output row j = original row perm[j] (GEGLU's [32 value | 32 gate] groups): gradient rows go back through the inverse
                assert len(mods) == 1 and x.p1 is None
                tmp = self.buf(n, k, torch.float32)
                self.full_wgrad(d, x.parts[0], tmp)
                inv = torch.empty_like(perm)
                inv[perm] = torch.arange(perm.numel())
                idx_w = self._idx(("rows", id(mod), n, k), lambda: (inv[:, None] * k + torch.arange(k)[None, :]).reshape(-1))
                ops.gather(tmp, idx_w, self.fgrad(w))
                self.pool.put(tmp)
                self.full_bias_grad(b, d, inv_perm=self._idx(("rows_b", id(mod), n), lambda: inv))
        self.drop(*x.parts)

    def full_conv_grads(self, mod, mode, dy):
        """dW (and db) of a 3x3 / strided / upsampled / temporal conv leaf; dy: its output gradient [M_out, N] (bf16, token-major)."""
        ops = self.ops
        x, info = self.full_take([mod])
        w, b = mod.weight, mod.bias
        n, cp = w.shape[0], w.shape[1]        # (the parameter's own channel counts: the operands may be zero-padded to 8)
        C = x.C
        taps = 3 if mode == nt.GEMM_TCONV3 else 9
        assert w.numel() == n * cp * taps and cp <= C and n <= dy.shape[1]
        rows = ops.im2col_rows(mode, x.n_img, x.h, x.w)
        assert rows == dy.shape[0], (rows, dy.shape)
        xcol = self.buf(rows, taps * C)
        ops.im2col(x.parts[0], x.p1, mode, x.n_img, x.h, x.w, info.get("frames", 0), xcol)
        tmp = self.buf(dy.shape[1], taps * C, torch.float32)
        self.full_wgrad(dy, xcol, tmp)
        self.pool.put(xcol)
        # parameter layout [N, C, k...] <- tap-major [N', taps, C']
        idx = self._idx(("conv", n, cp, C, taps), lambda: ((torch.arange(n)[:, None, None] * taps + torch.arange(taps)[None, None, :]) * C
                                                           + torch.arange(cp)[None, :, None]).reshape(-1))
        ops.gather(tmp, idx, self.fgrad(w))
        self.pool.put(tmp)
        self.full_bias_grad(b, dy)
        self.drop(*x.parts)

    def full_norm_grads(self, norm, x, dy, *, kind, silu=False, rows_per_unit=0, stats=None):
        """dgamma / dbeta of a GroupNorm (kind 0: ``x`` an Act, statistics of the forward) or LayerNorm (kind 1: ``x`` the rows)."""
        if not getattr(norm, "elementwise_affine", True) and kind == 1:
            return
        if norm.weight is None:
            return
        ops, pk = self.ops, self.pk
        x0, x1 = (x.parts[0], x.p1) if kind == 0 else (x, None)
        M, C = dy.shape
        ws = self.buf(1, max(ops.norm_affine_grad_ws_floats(M, M, C), 1), torch.float32)
        ops.norm_affine_grad(x0, x1, dy, kind=kind, sum_rows=M, ws=ws, dgamma=self.fgrad(norm.weight).view(1, C),
                             dbeta=self.fgrad(norm.bias).view(1, C), rows_per_unit=rows_per_unit,
                             groups=getattr(norm, "num_groups", 0), stats=stats, eps=getattr(norm, "eps", 0.0),
                             gamma=pk.f32(norm.weight), beta=pk.f32(norm.bias), silu=silu)
        self.pool.put(ws)

    def full_grads(self, params):
        """Gradients of ``params`` after ``backward``: views into ONE copy of the gradient arena (the engine's own buffers are overwritten by
        the next step; the copy lives as long as any of the returned tensors)."""
        plan = self._last
        if plan.get("fgrad_arena") is None:
            return [None for _ in params]
        flat = plan["fgrad_arena"].clone()
        out = []
        for p in params:
            if id(p) in plan["fgrads"]:
                o = plan["fgrad_off"][id(p)]
                out.append(flat[o:o + p.numel()].view(p.shape))
            else:
                out.append(None)
        return out
