"""LoRA branch and LoRA weight gradients of the student UNet on the native engine (the trainable part of the v1
distillation step: ``utils/lora.py:45-50,124-129,204-209`` forward, ``train_t2v_turbo_v1_lora.py:1190`` backward).

For an injected leaf  y = x (*) W + s * ((x (*) D) U^T)  with rank r (D = ``lora_down``, U = ``lora_up``, s = ``scale``,
(*) the leaf's gather: linear, 3x3 / strided / upsampled conv, (3,1,1) temporal conv) the engine runs

  forward    t = x (*) D            [M, rp]      same implicit-GEMM mode as the base leaf, N = rp (rank padded to 64)
             z = s t U^T (+ res)    [M, N]       K = rp
             y = x (*) W + z                      the frozen base leaf with z as its residual operand
  backward   dU = s dy^T t                        K = tokens: both operands transposed to K-contiguous bf16, split-K
             g  = s dy U             [M, rp]
             dx = adjoint(dy; W) + adjoint(g; D)  second term chained in as the residual of the first
             dD[r, tap, c] = sum_m G[m, tap, r] x[m, c],   G = adjoint(g; selection)   (the im2col matrix is never built:
                                                  the rank-r gradient is gathered to the INPUT grid instead — 9 rp columns
                                                  where im2col would have 9 Cin)

The frozen base weights are packed once.  The LoRA tensors change every optimizer step, so their operand layouts
(D forward / D data-gradient / U / U^T, bf16, zero-padded to rank 64) are re-laid from ONE flat fp32 copy of the trainable
parameters by ONE indexed-gather launch per step (``t2v_gather_f32``); one more gather carries every weight gradient from the
GEMM output layout into the flat gradient buffer (``dist.FlatGradSync``) that the all-reduce and ``optim.FlatAdamW`` work
on.  Leaves of the conditioning branch (time / fps / guidance MLPs, ``emb_layers``: M = B rows) stay in torch autograd; the
engine hands back d(loss)/d(emb_all), column sums of the ResBlock gradients taken as one more GEMM against a clip-indicator
row.  A train-mode student's dropouts (LoRA branch, temporal conv blocks) are counter-based masks (``t2v_dropout_bf16``):
a function of (step seed, site, element), applied in the forward and regenerated in the backward — not torch's random stream.

Status: dataflow verified on CPU against torch autograd (tests/test_unet_lora_grad_cpu.py) and on MI355X against autograd and the
reference's own LoRA gradients (tests/test_gpu_unet_grad.py)."""
import os

import torch
import torch.nn as nn

from . import native as nt
from .engine import is_lora_leaf

RP = 64  # rank granularity: K of a GEMM is a multiple of 64
# the gradient arena is all-reduced in this many pieces while the backward is still running (dist.FlatGradSync, N > 1)
ALLREDUCE_SEGMENTS = max(1, int(os.environ.get("T2V_ALLREDUCE_SEGMENTS", "8")))


def _pad(n, m):
    return (n + m - 1) // m * m


def _pad_to(t, shape):
    """int64 index tensor padded with -1 up to ``shape`` (trailing side of every dim)."""
    out = torch.full(shape, -1, dtype=torch.int64)
    out[tuple(slice(0, s) for s in t.shape)] = t
    return out


class LoraGroup:
    """1..n injected leaves reading the same input through the same gather mode (q/k/v of an attention = one group)."""
    saved = None


class LoraTrainMixin:
    lora_params = None
    # the token-contracted weight gradients by t2v_wgrad_tn on the token-major operands (no transposed copies); T2V_TN_WGRAD=0
    # selects t2v_transpose_pad_bf16 + split-K t2v_gemm instead (both validated on MI355X; the step is 313 -> 290 ms with this one)
    tn_wgrad = os.environ.get("T2V_TN_WGRAD", "1") == "1"
    # the LoRA branch's dropout as the epilogue of its up-projection GEMM (T2V_FUSE_DROPOUT=0: separate t2v_dropout_bf16 pass)
    fuse_dropout = os.environ.get("T2V_FUSE_DROPOUT", "1") == "1"
    # all weight gradients of a LoRA group (dU per leaf, per-clip column sums, dD per input part) as ONE t2v_wgrad_tn_group launch
    # pair once the rank-r gradient exists, instead of one t2v_wgrad_tn pair each (T2V_GROUP_WGRAD=0)
    group_wgrad = os.environ.get("T2V_GROUP_WGRAD", "1") == "1"
    # the LoRA branch's up-projection and dropout inside the base leaf's GEMM epilogue (t2v_gemm lora_* fields): two launches per
    # leaf group instead of 2 + leaves, the M x N up-projection never in memory.  Built at the end of round 3, timed in round 4
    # (same box, same process tree: 222.5 vs 227.0 ms per distillation step, student forward 47.3 vs 51.1 ms, 1 650 vs 2 150
    # launches; profiles/r04_distill_ab.jsonl) and the default since; T2V_LORA_EPILOGUE=0: one GEMM per leaf, as before.
    fuse_lora = os.environ.get("T2V_LORA_EPILOGUE", "1") == "1"

    # ---- binding ------------------------------------------------------------------------------------------------------
    def bind_lora(self, params):
        """``params``: the trainable LoRA tensors in flat-buffer order (``lora.lora_parameters(model)``)."""
        self.lora_params = list(params)
        self.lora_off, off = {}, 0
        for p in self.lora_params:
            self.lora_off[id(p)] = off
            off += p.numel()
        self.lora_numel = off
        self.lora_ids = set(self.lora_off)
        self.plans.clear()
        self.fingerprint = None
        self._engine_leaves = None

    @property
    def training_lora(self):
        return self.lora_params is not None

    def engine_leaves(self):
        """Injected leaves whose gradients the engine computes (token-row leaves; the B-row conditioning branch is torch's), in
        registration order.  Cached per binding: the module route asks on every call, and a walk in ``modules()`` order costs ms."""
        cached = getattr(self, "_engine_leaves", None)
        if cached is not None and all(is_lora_leaf(mod) for mod in cached[:4]):
            return cached
        self._engine_leaves = self._find_engine_leaves()
        return self._engine_leaves

    def _find_engine_leaves(self):
        m = self.model
        cond = set()
        for name in ("time_embed", "fps_embedding", "time_cond_proj", "motion_cond_proj", "combine_proj"):
            sub = getattr(m, name, None)
            if sub is not None:
                cond.update(id(x) for x in sub.modules())
        from .unet3d import ResBlock
        for mod in m.modules():
            if isinstance(mod, ResBlock):
                cond.update(id(x) for x in mod.emb_layers.modules())
        return [mod for mod in m.modules() if is_lora_leaf(mod) and id(mod) not in cond]

    def conditioning_parameters(self):
        mine = {id(p) for mod in self.engine_leaves() for p in (mod.lora_up.weight, mod.lora_down.weight)}
        return [p for p in self.lora_params if id(p) not in mine]

    # ---- arenas -------------------------------------------------------------------------------------------------------
    def _lora_begin(self):
        dev = self.device
        n_lp = n_e = 0
        for mod in self.engine_leaves():
            down, up = mod.lora_down.weight, mod.lora_up.weight
            if id(down) not in self.lora_off or id(up) not in self.lora_off:
                raise ValueError("a LoRA leaf's tensors are not in the list given to bind_lora")
            if not isinstance(mod.selector, nn.Identity):
                raise NotImplementedError("native LoRA training: a selector is set on an injected leaf")
            r, cin, n_out = down.shape[0], down.shape[1], up.shape[0]
            taps = down.numel() // (r * cin)
            rp, ce, npad = _pad(r, RP), _pad(cin, 64), _pad(n_out, 64)
            n_lp += 2 * rp * taps * ce + n_out * rp * (2 if self.fuse_lora else 1) + rp * npad + (taps * rp) ** 2 * (taps > 1)
            n_e += n_out * rp + taps * rp * ce
        self.lp = torch.zeros(n_lp, dtype=self.adt, device=dev)
        self.lp_idx = torch.full((n_lp,), -1, dtype=torch.int32, device=dev)
        self.lp_used = 0
        self.E = torch.zeros(n_e, dtype=torch.float32, device=dev)
        self.e_used = 0
        self.g_idx = torch.full((self.lora_numel,), -1, dtype=torch.int32, device=dev)
        self.src_flat = torch.empty(self.lora_numel + 1, dtype=torch.float32, device=dev)
        self.one_idx = self.lora_numel  # src_flat[-1] == 1: the constant entries of an operand (the selection packs) index it
        self._groups = {}
        self._packs_key = None
        self._refresh_src()

    def _refresh_src(self):
        ps = self.lora_params
        with torch.no_grad():
            first = ps[0]
            flat = None
            # optim.FlatAdamW re-homes the parameters as consecutive views of one buffer: take it as it is
            if all(p.is_contiguous() for p in ps) and first.dtype == torch.float32 and first.device == self.src_flat.device:
                base, ok = first.data_ptr(), True
                for p in ps:
                    if p.data_ptr() != base + 4 * self.lora_off[id(p)] or p.dtype != torch.float32:
                        ok = False
                        break
                if ok and first.untyped_storage().nbytes() - (first.storage_offset() * 4) >= 4 * self.lora_numel:
                    flat = torch.as_strided(first.detach(), (self.lora_numel,), (1,), first.storage_offset())
            if flat is not None:
                self.src_flat[:-1].copy_(flat)
            else:
                torch.cat([p.detach().reshape(-1).to(self.src_flat.device, torch.float32) for p in ps], out=self.src_flat[:-1])
            self.src_flat[-1] = 1.0

    def refresh_lora_packs(self):
        """Per step: flat parameters -> every bf16 operand layout, one launch.  Skipped while no LoRA tensor has changed since
        the last refresh (the target and the student forwards of one distillation step share the packs): every in-place update
        moves a version counter (FlatAdamW.step touches one on purpose), and a re-homed parameter moves its data pointer.
        Contract: a write that moves no version counter (through ``p.data``, or through a flat buffer the tensors are ``.data``
        views of — a restored ``flat_param``, an EMA or a custom optimizer on the flat buffer) must call
        ``invalidate_lora_packs()``.  ``FlatAdamW.step`` and ``update_ema_flat`` do NOT call it: they write through the flat buffer and
        then TOUCH one tensor's version counter on purpose (``add_(0.0)``), which is what this fingerprint sees — do not remove
        that touch on the strength of an invalidation hook that nothing registers."""
        fp = 0
        for p in self.lora_params:   # every version counter and every data pointer (re-homing of ANY tensor is seen)
            fp = (fp * 1000003 + p._version * 31 + (p.data_ptr() & 0xFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFF
        key = (fp, getattr(self, "_packs_epoch", 0))
        if key == getattr(self, "_packs_key", None):
            return
        self._packs_key = key
        self._refresh_src()
        if self.lp_used:
            self.ops.gather(self.src_flat, self.lp_idx[:self.lp_used], self.lp[:self.lp_used])

    def invalidate_lora_packs(self):
        """Force the next forward to re-gather the bf16 operand packs from the LoRA tensors (see ``refresh_lora_packs``)."""
        self._packs_epoch = getattr(self, "_packs_epoch", 0) + 1

    def _lp_alloc(self, idx2d):
        """Operand pack in the arena: records its gather indices and fills it (record time only: torch indexing)."""
        n = idx2d.numel()
        a = self.lp_used
        assert a + n <= self.lp.numel(), "LoRA pack arena overflow"
        self.lp_used = a + _pad(n, 8)
        di = idx2d.reshape(-1).to(self.device)
        self.lp_idx[a:a + n] = di.to(torch.int32)
        vals = torch.where(di >= 0, self.src_flat[di.clamp_min(0)], torch.zeros((), device=self.device))
        self.lp[a:a + n] = vals.to(self.adt)
        return self.lp[a:a + n].view(idx2d.shape)

    def _e_alloc(self, rows, cols):
        a = self.e_used
        assert a + rows * cols <= self.E.numel(), "LoRA gradient arena overflow"
        self.e_used = a + rows * cols
        return a, self.E[a:a + rows * cols].view(rows, cols)

    def sel_pack(self, taps, rp):
        """Selection weights of the gathered rank-r gradient: out[m, tap*rp + r] = g[m - offset(tap), r]  (tap flipped)."""
        def make():
            w = torch.zeros(taps, rp, taps, rp)
            for tap in range(taps):
                w[tap, :, taps - 1 - tap, :] = torch.eye(rp)
            return w.reshape(taps * rp, taps * rp).to(self.device, self.adt).contiguous()
        return self.pk._memo(("lora_sel", taps, rp), make)

    def clip_indicator(self, m_rows):
        """[B, Mp] bf16, 1 on the rows of clip b: A operand of the column-sum GEMM (d loss / d emb_all)."""
        def make():
            B = self.B
            mp = _pad(m_rows, 64)
            ind = torch.zeros(B, mp)
            per = m_rows // B
            for b in range(B):
                ind[b, b * per:(b + 1) * per] = 1
            return ind.to(self.device, self.adt).contiguous()
        return self.pk._memo(("clip_ind", m_rows, self.B), make)

    def clip_indicator_tok(self, m_rows):
        """[M, 8] bf16, column b = 1 on the rows of clip b: the token-major operand of the column sums for t2v_wgrad_tn."""
        def make():
            B = self.B
            assert B <= 8, "per-clip column sums through t2v_wgrad_tn: at most 8 clips per rank"
            ind = torch.zeros(m_rows, 8)
            per = m_rows // B
            for b in range(B):
                ind[b * per:(b + 1) * per, b] = 1
            return ind.to(self.device, self.adt).contiguous()
        return self.pk._memo(("clip_ind_tok", m_rows, self.B), make)

    # ---- groups -------------------------------------------------------------------------------------------------------
    def lgroup(self, mods, mode, perm=None):
        key = (mode,) + tuple(id(m) for m in mods)
        g = self._groups.get(key)
        if g is not None:
            return g
        taps = {nt.GEMM_LINEAR: 1, nt.GEMM_TCONV3: 3}.get(mode, 9)
        assert taps == 1 or len(mods) == 1
        g = LoraGroup()
        g.mods, g.mode, g.taps, g.n = mods, mode, taps, len(mods)
        g.N, g.npad, g.scale, g.Uf, g.UT, g.EU = [], [], [], [], [], []
        df_rows, db_cols = [], []
        r0 = mods[0].lora_down.weight.shape[0]
        g.rp = rp = _pad(r0, RP)
        g.cin = cin = mods[0].lora_down.weight.shape[1]
        g.ce = ce = _pad(cin, 64)
        g.e_lo = self.e_used
        e_d_off, g.ED = self._e_alloc((g.n if taps == 1 else taps) * rp, ce)
        for i, mod in enumerate(mods):
            down, up = mod.lora_down.weight, mod.lora_up.weight
            r, n_out = down.shape[0], up.shape[0]
            assert _pad(r, RP) == rp and down.shape[1] == cin
            idx_d = self.lora_off[id(down)] + torch.arange(down.numel()).view(down.shape)
            if taps == 9:
                t = idx_d.permute(0, 2, 3, 1).reshape(r, 9, cin)
            elif taps == 3:
                t = idx_d[:, :, :, 0, 0].permute(0, 2, 1)
            else:
                t = idx_d.reshape(r, 1, cin)
            t = _pad_to(t, (rp, taps, ce))
            df_rows.append(t.reshape(rp, taps * ce))
            db_cols.append(t.flip(1).permute(2, 1, 0).reshape(ce, taps * rp))   # [c][(tap', r)] = D[r][c][flipped tap']
            idx_u = self.lora_off[id(up)] + torch.arange(up.numel()).view(n_out, r)
            if perm is not None:
                idx_u = idx_u[perm]                                              # packed row j <- original row perm[j]
            uf = _pad_to(idx_u, (n_out, rp))
            g._uf_idx = getattr(g, "_uf_idx", []) + [uf]
            npad = _pad(n_out, 64)
            g.Uf.append(self._lp_alloc(uf))
            g.UT.append(self._lp_alloc(_pad_to(uf.t(), (rp, npad))))
            g.N.append(n_out)
            g.npad.append(npad)
            g.scale.append(float(mod.scale))
            # where each parameter element's gradient lands in the E arena
            e_u_off, eu = self._e_alloc(n_out, rp)
            g.EU.append(eu)
            e = e_u_off + torch.arange(n_out * rp).view(n_out, rp)[:, :r]
            if perm is not None:
                full = torch.empty_like(e)
                full[perm] = e
                e = full
            o = self.lora_off[id(up)]
            self.g_idx[o:o + up.numel()] = e.reshape(-1).to(self.device, torch.int32)
            if taps == 1:
                e = e_d_off + torch.arange(g.n * rp * ce).view(g.n, rp, ce)[i, :r, :cin]
            else:
                e = e_d_off + torch.arange(taps * rp * ce).view(taps, rp, ce)[:, :r, :cin].permute(1, 2, 0)
            o = self.lora_off[id(down)]
            self.g_idx[o:o + down.numel()] = e.reshape(-1).to(self.device, torch.int32)
        g.Df = self._lp_alloc(torch.cat(df_rows, dim=0))
        g.Ucat = self._lp_alloc(torch.cat(g._uf_idx, dim=0)) if self.fuse_lora else None   # [sum N][rp]: row n = leaf's up-projection row
        if taps > 1:
            # conv leaves: the data-gradient pack of D and the 0/1 selection pack of the gathered rank-r gradient stacked along N,
            # so ONE launch over g yields both (the selection's ones index the constant 1.0 behind the flat parameters)
            sel = torch.full((taps, rp, taps, rp), -1, dtype=torch.int64)
            for tap in range(taps):
                sel[tap, torch.arange(rp), taps - 1 - tap, torch.arange(rp)] = self.one_idx
            g.DbSel = self._lp_alloc(torch.cat([db_cols[0], sel.reshape(taps * rp, taps * rp)], dim=0))
            g.Db = g.DbSel[:ce]
        else:
            g.Db = self._lp_alloc(torch.cat(db_cols, dim=1))
        g.ntot = sum(g.N)
        g.e_hi = self.e_used
        self._groups[key] = g
        return g

    def saved_group(self, mods, mode):
        if not self.training_lora or mods is None:
            return None
        g = self._groups.get((mode,) + tuple(id(m) for m in mods))
        return g if g is not None and g.saved is not None else None

    # ---- forward: z = residual + sum_i s_i (x (*) D_i) U_i^T ----------------------------------------------------------------
    row_kind = "rows"  # how torch orders the rows of the current Linear leaves ("rows" | "temporal" | "ctx"): mask replay in tests

    def lora_t(self, grp, x, m_out, frames=0, kind_meta=None, kind=None):
        """The group's rank-r down-projections t = x (*) D [m_out, n * rp]; registers the dropout site, keeps x and t for the backward."""
        ops = self.ops
        nrp = grp.n * grp.rp
        t = self.buf(m_out, nrp)
        if grp.mode == nt.GEMM_LINEAR:
            ops.gemm(x.parts[0], grp.Df, t, M=m_out, N=nrp, a1=x.p1)
        else:
            ops.gemm(x.parts[0], grp.Df, t, M=m_out, N=nrp, a1=x.p1, mode=grp.mode, n_img=x.n_img, h=x.h, wd=x.w, frames=frames)
        grp.drop = self.drop_site([mm.dropout for mm in grp.mods], kind or (self.row_kind if grp.mode == nt.GEMM_LINEAR else "conv"), kind_meta)
        self.hold(*x.parts)
        grp.saved = (x, t)
        return t

    def lora_epilogue_args(self, grp):
        """kwargs that make the base leaf's ``ops.gemm`` add the group's LoRA branch in its epilogue (t2v_gemm lora_*), or None
        where the group does not fit that form (then: ``lora_up``)."""
        if not self.fuse_lora or grp.Ucat is None or grp.rp != 64 or len(set(grp.N)) != 1 or grp.N[0] % 32 or len(set(grp.scale)) != 1:
            return None
        _, t = grp.saved
        kw = {"lora": (t, grp.Ucat, grp.N[0], grp.scale[0])}
        if grp.drop:
            kw["dropout"] = (grp.drop[0], self.seed_t, grp.drop[1], grp.ntot, 0)
        return kw

    def lora_up(self, grp, m_out, residual=None):
        """z = residual + sum_i s_i dropout(t_i U_i^T) as launches of their own.  Returns (z buffer to release, z view [m_out, sum N])."""
        ops = self.ops
        _, t = grp.saved
        zf = self.buf(m_out, _pad(grp.ntot, 8))
        z = zf[:, :grp.ntot]
        # train mode: dropout(up(down(x))) * scale (utils/lora.py:45-50), then the leaf's own residual.  The mask is the
        # up-projection's own epilogue (t2v_gemm dropout fields) instead of a separate read-modify-write pass over z
        # (442 launches and 23 GB per student forward); ``fuse_dropout = False`` keeps the two-kernel form (tests compare both).
        fuse = bool(grp.drop) and self.fuse_dropout and grp.ntot % 4 == 0 and all(n % 4 == 0 for n in grp.N)
        c0 = 0
        for i in range(grp.n):
            res = None if (residual is None or (grp.drop and not fuse)) else residual[:, c0:c0 + grp.N[i]]
            drop = (grp.drop[0], self.seed_t, grp.drop[1], grp.ntot, c0) if fuse else None
            ops.gemm(t[:, i * grp.rp:(i + 1) * grp.rp], grp.Uf[i], z[:, c0:c0 + grp.N[i]], M=m_out, N=grp.N[i],
                     alpha=grp.scale[i], residual=res, dropout=drop)
            c0 += grp.N[i]
        if grp.drop and not fuse:
            ops.dropout(z, None if residual is None else residual[:, :grp.ntot], z, grp.ntot, grp.drop[0], self.seed_t, grp.drop[1])
        return zf, z

    def lora_z(self, grp, x, m_out, residual=None, frames=0, kind_meta=None, kind=None):
        """x: Act (1 or 2 parts, channels padded to the group's ce).  Returns (z buffer to release, z view [m_out, sum N])."""
        self.lora_t(grp, x, m_out, frames, kind_meta, kind)
        return self.lora_up(grp, m_out, residual)

    # ---- backward ---------------------------------------------------------------------------------------------------------
    def tposed(self, src, rows, cols, batch=1, in_stride=0):
        """[batch][rows][cols] -> [batch][cols][rows padded to 64 with zeros] (K-contiguous GEMM operand)."""
        rp = _pad(rows, 64)
        dst = self.buf(batch * cols, rp)
        self.ops.transpose_pad(src, rows, cols, dst, batch=batch, in_stride=in_stride, out_stride=cols * rp)
        return dst

    @staticmethod
    def split_for(m, n, k):
        """Split-K factor of a weight-gradient GEMM: a handful of output tiles, K = tokens."""
        tiles = ((m + 127) // 128) * ((n + 127) // 128)
        return max(1, min(448 // tiles, k // 512, 64))

    def lora_wgrad(self, grp, dy, colsum=None):
        """dU of every leaf of the group (+ optional per-clip column sums of dy) and g = s dy U.  dy: [M, sum npad]."""
        ops = self.ops
        x, t = grp.saved
        m = dy.shape[0]
        mp = _pad(m, 64)
        dy_plain = dy
        if grp.drop:  # the LoRA branch saw dy through the forward's mask; the base leaf and the row vector see dy itself
            dym = self.buf(m, dy.shape[1])
            if dy.shape[1] != grp.ntot:
                ops.fill_zero(dym)
            ops.dropout(dy, None, dym, grp.ntot, grp.drop[0], self.seed_t, grp.drop[1])
            dy = dym
        g = self.buf(m, grp.n * grp.rp)
        if self.tn_wgrad:  # token-contracted kernel on the token-major operands themselves: no transposed copies
            grouped = self.group_wgrad and hasattr(ops, "wgrad_tn_group")
            pending = []
            c0 = 0
            for i in range(grp.n):
                n_out, rp = grp.N[i], grp.rp
                prob = (dy[:, c0:c0 + n_out], t[:, i * rp:(i + 1) * rp], grp.EU[i], grp.scale[i])
                if grouped:
                    pending.append(prob)
                else:
                    ops.wgrad_tn(prob[0], prob[1], prob[2], alpha=prob[3])
                ops.gemm(dy[:, c0:c0 + grp.npad[i]], grp.UT[i], g[:, i * rp:(i + 1) * rp], M=m, N=rp, alpha=grp.scale[i])
                c0 += grp.npad[i]
            if colsum is not None:
                ind = self.clip_indicator_tok(m)
                prob = (ind[:, :self.B], dy_plain[:, :grp.N[0]], colsum, 1.0)
                if grouped:
                    pending.append(prob)
                else:
                    ops.wgrad_tn(prob[0], prob[1], prob[2])
            if grouped:   # launched together with dD by lora_wgrad_down (every caller goes there next); dy / t stay until then
                grp.pending = (pending, [dy] if grp.drop else [], t)
                return g
            if grp.drop:
                self.pool.put(dy)
            self.pool.put(t)
            return g
        dyT = self.tposed(dy, m, dy.shape[1])
        tT = self.tposed(t, m, t.shape[1])
        c0 = 0
        for i in range(grp.n):
            n_out, rp = grp.N[i], grp.rp
            ops.gemm(dyT[c0:c0 + n_out], tT[i * rp:(i + 1) * rp], grp.EU[i], M=n_out, N=rp, alpha=grp.scale[i],
                     split_k=self.split_for(n_out, rp, mp))
            ops.gemm(dy[:, c0:c0 + grp.npad[i]], grp.UT[i], g[:, i * rp:(i + 1) * rp], M=m, N=rp, alpha=grp.scale[i])
            c0 += grp.npad[i]
        if colsum is not None:
            ind = self.clip_indicator(m)
            src = dyT
            if grp.drop:
                src = self.tposed(dy_plain, m, dy_plain.shape[1])
            ops.gemm(ind, src[:grp.N[0]], colsum, M=ind.shape[0], N=grp.N[0], split_k=self.split_for(ind.shape[0], grp.N[0], mp))
            if grp.drop:
                self.pool.put(src)
        if grp.drop:
            self.pool.put(dy)
        self.pool.put(dyT, tT, t)
        return g

    def lora_wgrad_down(self, grp, G):
        """dD from the rank-r gradient gathered to the input grid: G [M_in, rows of ED] against the saved input."""
        ops = self.ops
        x, _ = grp.saved
        m_in = x.M
        mp = _pad(m_in, 64)
        if self.tn_wgrad:
            pending, masked, t = getattr(grp, "pending", None) or ([], [], None)
            grp.pending = None
            c0 = 0
            for part in x.parts:
                c = part.shape[1]
                if t is not None:
                    pending.append((G, part, grp.ED[:, c0:c0 + c], 1.0))
                else:
                    ops.wgrad_tn(G, part, grp.ED[:, c0:c0 + c])
                c0 += c
            if t is not None:
                ops.wgrad_tn_group(pending)
                self.pool.put(*masked, t)
            self.drop(*x.parts)
            grp.saved = None
            self._group_finished(grp)
            return
        GT = self.tposed(G, m_in, G.shape[1])
        c0 = 0
        for part in x.parts:
            c = part.shape[1]
            xT = self.tposed(part, m_in, c)
            ops.gemm(GT, xT, grp.ED[:, c0:c0 + c], M=GT.shape[0], N=c, split_k=self.split_for(GT.shape[0], c, mp))
            self.pool.put(xT)
            c0 += c
        self.pool.put(GT)
        self.drop(*x.parts)
        grp.saved = None
        self._group_finished(grp)

    def lora_grads_into(self, flat_grad, accumulate=True, alpha=1.0):
        """E arena -> flat gradient buffer in parameter layout (one launch).  Conditioning-branch tensors are not touched
        when accumulating (their gradient comes from torch), and zeroed otherwise."""
        self.ops.gather(self.E, self.g_idx, flat_grad, alpha=alpha, accumulate=accumulate)

    # ---- gradient exchange overlapped with the backward (train_t2v_turbo_v1_lora.py:1190 under DDP: buckets reduced as they fill) ---
    # The weight gradients land in the arena E in GEMM output layout, groups in forward order, and the backward completes them from
    # the top of the arena down.  The arena layout is the same on every rank, so it is all-reduced AS IT IS, in ALLREDUCE_SEGMENTS
    # contiguous pieces, each as soon as every group that touches it has its dU and dD: the recorded backward list carries a marker
    # entry per piece (a host call between two launches) that issues ``all_reduce(E[lo:hi], async_op=True)`` — the process group
    # orders it after the launches enqueued so far and runs it on its own stream.  ``backward`` waits for the handles and the ONE
    # gather launch that re-lays E into parameter order applies the 1 / world factor.  The conditioning branch's few tensors
    # (torch autograd, after the engine's backward) are reduced by ``FlatGradSync.all_reduce_mean`` as a subset.
    _overlap = None

    def _seg_bounds(self):
        k = ALLREDUCE_SEGMENTS
        return [0 if i == 0 else self.e_used if i == k else (self.e_used * i // k) // 1024 * 1024 for i in range(k + 1)]

    def _seg_reset(self):
        self._seg_done, self._seg_emitted = set(), 0
        self._seg_order = sorted(self._groups.values(), key=lambda g: -g.e_lo)

    def _group_finished(self, grp):
        if getattr(self, "_seg_done", None) is None:
            return
        self._seg_done.add(id(grp))
        tail = self.e_used
        for g in self._seg_order:       # the arena is complete from `tail` up once every group allocated above it is done
            if id(g) not in self._seg_done:
                break
            tail = g.e_lo
        self._seg_emit_down_to(tail)

    def _seg_emit_down_to(self, tail):
        k, b = ALLREDUCE_SEGMENTS, self._seg_bounds()
        while self._seg_emitted < k and b[k - 1 - self._seg_emitted] >= tail:
            j = k - 1 - self._seg_emitted
            self._seg_emitted += 1
            self._segment_hook(j, b[j], b[j + 1])
            rec = getattr(self.ops, "record_host_call", None)
            if rec is not None:         # replayed between the launches on either side of it
                rec(self._segment_hook, (j, b[j], b[j + 1]), "allreduce_segment")

    def _segment_hook(self, j, lo, hi, stream=None):
        if self._overlap is not None and hi > lo:
            import torch.distributed as dist
            self._handles.append(dist.all_reduce(self.E[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        return 0

    def conditioning_index(self):
        """int64 positions (flat-buffer order) of the LoRA tensors whose gradients torch computes (the B-row conditioning branch)."""
        idx = getattr(self, "_cond_idx", None)
        if idx is None or idx[0] is not self.lora_params:
            parts = [torch.arange(self.lora_off[id(p)], self.lora_off[id(p)] + p.numel()) for p in self.conditioning_parameters()]
            t = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64)
            self._cond_idx = idx = (self.lora_params, t.to(self.E.device))
        return idx[1]
