"""Torch emulation of the HIP op backend (``t2v_turbo_amd.native.HipOps``) — TEST ONLY.

Implements the *documented semantics* of every C-ABI entry point (include/t2v_hip.h) with plain
torch ops in fp32, consuming exactly the same tensor views / strides / packed weight layouts the
engine hands to the real kernels.  It lets the CPU test-suite check the engine's recorded dataflow
(layouts, weight packing, virtual concat, buffer reuse, batching strides) against the oracle
without a GPU, and doubles as the per-op reference for the GPU kernel tests.  It is never imported
by the product package.
"""
import math

import torch
import torch.nn.functional as F

from t2v_turbo_amd import native as nt


def _strided(t, rows, cols, ld, offset_elems):
    return torch.as_strided(t, (rows, cols), (ld, 1), t.storage_offset() + offset_elems)


class EmuOps:
    is_native = False

    def __init__(self, act_dtype=torch.float32, strict=False):
        self.act_dtype = act_dtype
        self.strict = strict  # also enforce the device-side operand alignment rules of t2v_gemm (engine dataflow tests)
        self.calls = []

    def init(self):
        pass

    def _log(self, name):
        self.calls.append(name)

    # ------------------------------------------------------------------------------------ gemm
    def gemm(self, a0, w, out, *, M, N, a1=None, mode=nt.GEMM_LINEAR, n_img=0, h=0, wd=0, frames=0, bias=None,
             rowvec=None, rowvec_div=0, residual=None, act=nt.ACT_NONE, alpha=1.0, batch=1, batch_inner=1,
             a_strides=(0, 0), w_strides=(0, 0), o_strides=(0, 0), tile_cfg=0, split_k=0, dropout=None, ln=None,
             rowstat=None, colstat=None, lnf=None, lora=None):
        self._log("gemm")
        if rowstat is not None or colstat is not None or lnf is not None or lora is not None:
            assert self.gemm_fuse_supported(a0, w, out, M=M, N=N, a1=a1, mode=mode, bias=bias, rowvec=rowvec, residual=residual, act=act,
                                            alpha=alpha, batch=batch, dropout=dropout, ln=ln, rowstat=rowstat, colstat=colstat, lnf=lnf,
                                            lora=lora)
        if ln is not None:
            assert batch == 1 and act == nt.ACT_NONE and alpha == 1.0, "LN output: no batch / activation / alpha (the device kernel: N == 320)"
        # device-side argument rules (csrc/gemm.hip, t2v_gemm): operand row strides, batch strides and base addresses
        if self.strict:
            assert a0.stride(0) % 8 == 0 and w.stride(0) % 8 == 0 and (a1 is None or a1.stride(0) % 8 == 0), "lda/ldw % 8"
            assert all(v % 8 == 0 for v in tuple(a_strides) + tuple(w_strides)), "batch strides % 8"
            assert a0.storage_offset() % 8 == 0 and w.storage_offset() % 8 == 0, "operand base must be 16-byte aligned"
            assert residual is None or residual.dtype == self.act_dtype, "residual is an activation-dtype tensor"
        c0 = a0.shape[1]
        c1 = 0 if a1 is None else a1.shape[1]
        cin = c0 + c1
        assert c0 % 64 == 0 and c1 % 64 == 0
        taps = {nt.GEMM_LINEAR: 1, nt.GEMM_TCONV3: 3}.get(mode, 9)
        K = taps * cin
        n_out = N // 2 if act == nt.ACT_GEGLU else N
        for z in range(batch):
            z0, z1 = z // batch_inner, z % batch_inner
            a_off = z0 * a_strides[0] + z1 * a_strides[1]
            w_off = z0 * w_strides[0] + z1 * w_strides[1]
            o_off = z0 * o_strides[0] + z1 * o_strides[1]
            wz = _strided(w, N, K, w.stride(0), w_off).float()
            if mode == nt.GEMM_LINEAR:
                rows = M
            else:
                rows = n_img * h * wd
            src = _strided(a0, rows, c0, a0.stride(0), a_off).float()
            if a1 is not None:
                src = torch.cat([src, _strided(a1, rows, c1, a1.stride(0), a_off).float()], dim=1)
            if mode == nt.GEMM_LINEAR:
                y = src @ wz.t()
            elif mode == nt.GEMM_TCONV3:
                b = n_img // frames
                x5 = src.reshape(b, frames, h * wd, cin).permute(0, 3, 1, 2)  # b c f p
                wk = wz.reshape(N, 3, cin).permute(0, 2, 1)[..., None]      # N c 3 1
                y = F.conv2d(x5, wk, padding=(1, 0)).permute(0, 2, 3, 1).reshape(-1, N)
            else:
                x4 = src.reshape(n_img, h, wd, cin).permute(0, 3, 1, 2)
                wk = wz.reshape(N, 3, 3, cin).permute(0, 3, 1, 2)
                if mode == nt.GEMM_CONV3X3:
                    y = F.conv2d(x4, wk, padding=1)
                elif mode == nt.GEMM_CONV3X3_S2:
                    y = F.conv2d(x4, wk, stride=2, padding=1)
                elif mode == nt.GEMM_CONV3X3_S2_PAD01:
                    y = F.conv2d(F.pad(x4, (0, 1, 0, 1)), wk, stride=2)
                elif mode == nt.GEMM_CONV3X3_UP2:
                    y = F.conv2d(F.interpolate(x4, scale_factor=2, mode="nearest"), wk, padding=1)
                else:
                    raise ValueError(mode)
                y = y.permute(0, 2, 3, 1).reshape(-1, N)
            assert y.shape[0] == M, (y.shape, M)
            y = y * alpha
            if lnf is not None:  # LayerNorm folded into the GEMM: raw rows in A, W diag(gamma) in W, statistics from the producer
                stats, eps_ln, s_vec = lnf
                nb = cin // 32
                st = stats.float()[:, :2 * nb].reshape(M, nb, 2)
                mean = st[:, :, 0].sum(dim=1) / cin
                var = (st[:, :, 1].sum(dim=1) / cin - mean * mean).clamp_min(0.0)
                rstd = 1.0 / torch.sqrt(var + eps_ln)
                y = rstd[:, None] * (y - mean[:, None] * s_vec.float()[None, :N])
            if bias is not None:
                y = y + bias.float()[None, :N]
            if lora is not None:  # the LoRA branch of the leaf(s) in the epilogue: + scale * dropout(t_leaf u^T), the mask on that product only
                t_l, u_l, n_leaf, sc = lora
                z = torch.cat([t_l.float()[:, 64 * l:64 * l + 64] @ u_l.float()[l * n_leaf:(l + 1) * n_leaf, :64].t()
                               for l in range(N // n_leaf)], dim=1)
                if dropout is not None and dropout[0] > 0:
                    p_drop, seed_t, site, ncols, col0 = dropout
                    keep = self.dropout_keep(int(seed_t.reshape(-1)[0]), site, M, ncols, p_drop)
                    if getattr(self, "masks", None) is not None:
                        self.masks[site] = keep
                    z = torch.where(keep[:, col0:col0 + N], z * nt.dropout_inv_keep(p_drop), torch.zeros(()))
                y = y + sc * z
            elif dropout is not None and dropout[0] > 0:  # the dropout epilogue of t2v_gemm: the mask of dropout() on its column block
                p_drop, seed_t, site, ncols, col0 = dropout
                assert act == nt.ACT_NONE and batch == 1 and ncols % 2 == 0 and col0 % 2 == 0 and col0 + N <= ncols
                keep = self.dropout_keep(int(seed_t.reshape(-1)[0]), site, M, ncols, p_drop)
                if getattr(self, "masks", None) is not None:
                    self.masks[site] = keep
                y = torch.where(keep[:, col0:col0 + N], y * nt.dropout_inv_keep(p_drop), torch.zeros(()))
            if act == nt.ACT_GEGLU:
                g = y.reshape(M, N // 64, 2, 32)
                y = (g[:, :, 0] * F.gelu(g[:, :, 1])).reshape(M, N // 2)
            if rowvec is not None:
                idx = torch.arange(M) // rowvec_div
                y = y + rowvec.float()[idx, :n_out]
            if residual is not None:
                y = y + _strided(residual, M, n_out, residual.stride(0), o_off).float()
            if act == nt.ACT_SILU:
                y = F.silu(y)
            _strided(out, M, n_out, out.stride(0), o_off).copy_(y.to(out.dtype))
            if rowstat is not None:  # (sum, sumsq) of the fp32 epilogue values per row and 32-column block
                yb = y.reshape(M, N // 32, 32)
                rowstat[:, :2 * (N // 32)] = torch.stack([yb.sum(dim=2), (yb * yb).sum(dim=2)], dim=2).reshape(M, -1)
            if colstat is not None:  # (sum, sumsq) per column of every 32-row slab, of the values as stored (output dtype)
                yo = y.to(out.dtype).float().reshape(M // 32, 32, n_out)
                colstat.view(M // 32, n_out, 2).copy_(torch.stack([yo.sum(dim=1), (yo * yo).sum(dim=1)], dim=2))
            if ln is not None:  # LayerNorm of the fp32 epilogue values (before they are rounded to the output dtype), second output
                gamma, beta, eps, out2 = ln
                out2[:, :N] = F.layer_norm(y, (N,), gamma.float(), beta.float(), eps).to(out2.dtype)

    # ---- t2v_linear_pr: short-K Linear on the fragment pack (csrc/linear_pr.hip) ---------------------------------------------------
    def linear_pr_supported(self, a0, wp, out, *, M, N, a1=None, mode=nt.GEMM_LINEAR, bias=None, rowvec=None, residual=None,
                            act=nt.ACT_NONE, alpha=1.0, batch=1, split_k=0, dropout=None, ln=None, rowstat=None, colstat=None, lnf=None,
                            lora=None, ln_in=None, gn_in=None, **_):
        """Mirror of lpr_prepare (csrc/linear_pr.hip): 0 not taken, 1 taken."""
        if ln_in is not None and residual is not None:
            return 0
        if gn_in is not None:
            bm = 160 if a0.shape[1] == 320 else 96
            if residual is not None or ln_in is not None or act != nt.ACT_NONE or gn_in[1] % bm or M % gn_in[1]:
                return 0
        if mode != nt.GEMM_LINEAR or a1 is not None or batch > 1 or alpha != 1.0 or split_k > 1 or out.dtype not in (self.act_dtype, torch.bfloat16):
            return 0
        if any(v is not None for v in (dropout, ln, rowstat, colstat, lnf, lora, rowvec)) or act not in (nt.ACT_NONE, nt.ACT_GEGLU):
            return 0
        if a0.shape[1] not in (320, 512, 640) or N % 64 or a0.stride(0) % 8 or out.stride(0) % 8:
            return 0
        if a0.shape[1] == 512 and (residual is not None or gn_in is not None):
            return 0
        if residual is not None and (act == nt.ACT_GEGLU or residual.stride(0) % 8 or M % 32):
            return 0
        return 1

    def linear_pr(self, a0, wp, out, **kw):
        self._log("linear_pr")
        assert self.linear_pr_supported(a0, wp, out, **kw), "t2v_linear_pr would refuse this launch"
        kw.pop("tile_cfg", None)
        gn_in = kw.pop("gn_in", None)
        if gn_in is not None:   # GroupNorm affine of the rows in the panel fill: rows * coef[u][0] + coef[u][1]
            coef, rpu = gn_in
            K = a0.shape[1]
            cf = coef.float().view(-1, 2, K)
            a0 = (a0.float().view(-1, rpu, K) * cf[:, None, 0, :] + cf[:, None, 1, :]).reshape(-1, K).to(a0.dtype)
        ln_in = kw.pop("ln_in", None)
        if ln_in is not None:   # LayerNorm of the rows in the panel fill: normalised rows rounded to the activation dtype, as t2v_layernorm writes them
            gamma, beta, eps = ln_in
            a0 = F.layer_norm(a0.float(), (a0.shape[1],), gamma.float(), beta.float(), eps).to(a0.dtype)
        self.gemm(a0, nt.unpack_linear_pr(wp[:kw["N"]]), out, **kw)

    # ---- t2v_conv_halo: the same 3x3 convolution on the slab-major weight pack (csrc/conv_halo.hip) ----------------------------
    HALO_TILES = ((10, 32, 160, 4), (10, 32, 80, 4), (10, 16, 80, 8), (5, 32, 80, 8), (10, 32, 128, 4))   # (rows, columns, channels, pairs per stage pair)

    def conv_halo_supported(self, a0, w, out, *, M, N, a1=None, mode=nt.GEMM_LINEAR, n_img=0, h=0, wd=0, frames=0, bias=None,
                            rowvec=None, rowvec_div=0, residual=None, act=nt.ACT_NONE, alpha=1.0, batch=1, tile_cfg=0, split_k=0,
                            dropout=None, ln=None, rowstat=None, colstat=None, lnf=None, lora=None, **_):
        """Mirror of halo_prepare (csrc/conv_halo.hip): 0 not taken, 1 taken."""
        if mode not in (nt.GEMM_CONV3X3, nt.GEMM_CONV3X3_UP2) or batch > 1 or alpha != 1.0 or out.dtype not in (self.act_dtype, torch.bfloat16) or split_k > 1:
            return 0
        ups = 1 if mode == nt.GEMM_CONV3X3_UP2 else 0
        if dropout is not None or ln is not None or rowstat is not None or lnf is not None or lora is not None or act not in (nt.ACT_NONE, nt.ACT_SILU):
            return 0
        c0, c1 = a0.shape[1], (0 if a1 is None else a1.shape[1])
        if c0 % 64 or c1 % 64 or N % 16 or (rowvec is not None and (rowvec_div <= 0 or rowvec_div % ((h << ups) * (wd << ups)))):
            return 0
        assert w.shape[1] >= nt.conv_halo_pack_cols(c0 + c1), "slab-major pack narrower than conv_halo_pack_cols"
        U, V = h << ups, wd << ups
        pick = tile_cfg - 39 if 40 <= tile_cfg < 40 + len(self.HALO_TILES) else 0
        best, best_id = -1.0, 0
        for i, (S, fx, bn, _) in enumerate(self.HALO_TILES, start=1):
            if pick and i != pick:
                continue
            if V % fx or (fx == 16 and V >= 32) or S >= 2 * U:
                continue
            tiles = n_img * ((U + S - 1) // S) * (V // fx) * ((N + bn - 1) // bn)
            rounds = (tiles + 255) // 256
            eff = tiles / (256.0 * rounds) * U / (((U + S - 1) // S) * S) * N / (((N + bn - 1) // bn) * bn)
            score = 2.0 + S * fx * bn * 1e-6 if eff >= 0.85 else eff
            if score > best:
                best, best_id = score, i
        if not best_id:
            return 0
        S, fx, _, _ = self.HALO_TILES[best_id - 1]
        if colstat is not None:
            if U % S or M % 32 or N % 2:
                return 0
            if fx == 16 and (V != 16 or S % 2 or (U * V) % 32):
                return 0
        return 1

    def conv_halo(self, a0, w, out, **kw):
        self._log("conv_halo")
        assert self.conv_halo_supported(a0, w, out, **kw), "t2v_conv_halo would refuse this launch"
        cin = a0.shape[1] + (0 if kw.get("a1") is None else kw["a1"].shape[1])
        assert not w[:, 9 * cin:].any(), "the padding columns of a slab-major pack must be zero"
        colstat = kw.pop("colstat", None)
        kw.pop("tile_cfg", None)
        self.gemm(a0, nt.unpack_conv_slab(w, cin), out, **kw)   # (residual and activation in fp32 before the one rounding, as the kernel)
        if colstat is not None:
            M, N = kw["M"], kw["N"]
            yo = out[:M, :N].float().reshape(M // 32, 32, N)
            colstat.view(M // 32, N, 2).copy_(torch.stack([yo.sum(dim=1), (yo * yo).sum(dim=1)], dim=2))

    def gemm_fuse_supported(self, a0, w, out, *, M, N, a1=None, mode=nt.GEMM_LINEAR, bias=None, rowvec=None, residual=None,
                            act=nt.ACT_NONE, alpha=1.0, batch=1, dropout=None, ln=None, rowstat=None, colstat=None, lnf=None,
                            lora=None, **_):
        """The argument rules of t2v_gemm_fuse_supported (csrc/gemm.hip) that do not depend on the tile: the emulated backend
        takes the fused form wherever the descriptor allows it, so that the CPU suite covers the dataflow at every width."""
        if lora is not None:   # LoRA epilogue, alone or with column statistics; its dropout masks the LoRA product
            t_l, u_l, n_leaf, _sc = lora
            ok = (rowstat is None and lnf is None and batch == 1 and ln is None and alpha == 1.0 and act == nt.ACT_NONE and
                  out.dtype == self.act_dtype and N % 16 == 0 and n_leaf % 32 == 0 and N % n_leaf == 0 and
                  t_l.shape[1] >= 64 * (N // n_leaf) and u_l.shape[0] == N and u_l.shape[1] >= 64)
            return ok and (colstat is None or M % 32 == 0)
        n_req = sum(x is not None for x in (rowstat, colstat, lnf))
        if n_req != 1 or batch != 1 or ln is not None or alpha != 1.0 or (dropout is not None and dropout[0] > 0):
            return False
        n_out = N // 2 if act == nt.ACT_GEGLU else N
        if out.dtype != self.act_dtype or n_out % 16 or N % 16:
            return False
        if rowstat is not None:
            return N % 32 == 0 and act == nt.ACT_NONE and rowstat.stride(0) >= N // 16 and rowstat.stride(0) % 4 == 0
        if colstat is not None:
            return M % 32 == 0 and act != nt.ACT_GEGLU
        cin = a0.shape[1]
        return (mode == nt.GEMM_LINEAR and a1 is None and bias is not None and residual is None and rowvec is None and cin % 64 == 0
                and cin <= 1280 and act != nt.ACT_SILU and lnf[0].stride(0) >= cin // 16 and lnf[0].stride(0) % 4 == 0)

    def gemm_plan(self, a0, w, out, **_):
        return 0, 1   # (the emulation has no tiles and never splits K: the fused forms are taken wherever the arguments allow)

    def ffn_fused_supported(self, C):
        return C % 32 == 0   # (the device kernel: C = 320 and 64; the emulation takes any width the layout allows)

    def ffn_fused(self, x, w1p, b1p, w2p, b2, eps, out):
        """out = x + W2 (value * gelu(gate)) + b2 with [value | gate] = W1' LayerNorm_noaffine(x) + b1' — decoded from the PACKED
        operands by the layout rules of include/t2v_hip.h (an independent restatement of native.ffn_pack's index maps)."""
        self._log("ffn_fused")
        M, C = x.shape
        nch, four, ks, _, _ = w1p.shape
        inner, rt = 32 * nch, C // 16
        assert four == 4 and ks == C // 32 and w2p.shape == (nch, rt, 64, 8) and b1p.shape == (nch, 4, 16)
        w1 = torch.zeros(2 * inner, C)
        b1 = torch.zeros(2 * inner)
        w2 = torch.zeros(C, inner)
        lanes = torch.arange(64)
        for jc in range(nch):    # piece by piece, by the layout rules of the header (vectorised over the 64 lanes of a piece)
            for T in range(4):
                r0 = (T % 2) * inner + 32 * jc + 16 * (T // 2)
                b1[r0:r0 + 16] = b1p[jc, T].float()
                rows = r0 + (lanes & 15)
                for s_ in range(ks):
                    cols = 32 * s_ + 8 * (lanes >> 4)
                    w1[rows[:, None], cols[:, None] + torch.arange(8)[None, :]] = w1p[jc, T, s_].float()
            for t in range(rt):
                orow = 16 * t + (lanes & 15)
                ha = 32 * jc + 4 * (lanes >> 4)
                w2[orow[:, None], ha[:, None] + torch.arange(4)[None, :]] = w2p[jc, t, :, :4].float()
                w2[orow[:, None], 16 + ha[:, None] + torch.arange(4)[None, :]] = w2p[jc, t, :, 4:].float()
        xf = x.float()
        xn = F.layer_norm(xf, (C,), None, None, eps)
        hcat = xn @ w1.t() + b1
        g = hcat[:, :inner] * F.gelu(hcat[:, inner:])
        out.copy_((xf + g @ w2.t() + b2.float()).to(out.dtype))

    def conv_small(self, x, n_img, h, w, wgt, bias, out):
        self._log("conv_small")
        cin, cout = x.shape[1], out.shape[1]
        x4 = x.float().reshape(n_img, h, w, cin).permute(0, 3, 1, 2)
        wk = wgt.float().reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)
        y = F.conv2d(x4, wk, None if bias is None else bias.float(), padding=1)
        out.copy_(y.permute(0, 2, 3, 1).reshape(-1, cout).to(out.dtype))

    @staticmethod
    def conv_small_cout_supported(w, cin, cout):
        return w > 0 and w % 4 == 0 and cin > 0 and cin % 8 == 0 and 1 <= cout <= 4 and 9 * cin * 16 <= 150 * 1024

    def conv_small_cout(self, x, n_img, h, w, wgt, bias, out):
        self._log("conv_small_cout")
        assert self.conv_small_cout_supported(w, x.shape[1], out.shape[1])
        self.conv_small(x, n_img, h, w, wgt, bias, out)
        self.calls.pop()

    # ------------------------------------------------------------------------------------ norms
    def gn_ws_floats(self, n_units, rows_per_unit, groups=32):
        return 8

    @staticmethod
    def _cat(x0, x1):
        return x0.float() if x1 is None else torch.cat([x0.float(), x1.float()], dim=1)

    def gn_stats(self, x0, x1, n_units, rows_per_unit, eps, ws, stats, groups=32):
        self._log("gn_stats")
        x = self._cat(x0, x1)
        C = x.shape[1]
        xg = x.reshape(n_units, rows_per_unit, groups, C // groups).permute(0, 2, 1, 3).reshape(n_units, groups, -1)
        mean = xg.mean(dim=2)
        var = xg.var(dim=2, unbiased=False)
        stats.copy_(torch.stack([mean, 1.0 / torch.sqrt(var + eps)], dim=2).reshape(n_units, groups * 2))

    def gn_apply(self, x0, x1, n_units, rows_per_unit, stats, gamma, beta, silu, out, groups=32):
        self._log("gn_apply")
        x = self._cat(x0, x1)
        C = x.shape[1]
        st = stats.reshape(n_units, groups, 2)
        mean = st[:, :, 0].repeat_interleave(C // groups, dim=1)[:, None, :]
        rstd = st[:, :, 1].repeat_interleave(C // groups, dim=1)[:, None, :]
        y = (x.reshape(n_units, rows_per_unit, C) - mean) * rstd * gamma.float() + beta.float()
        if silu:
            y = F.silu(y)
        out.copy_(y.reshape(-1, C).to(out.dtype))

    def group_norm_ws_floats(self, n_units, rows_per_unit, groups, channels):
        return 8

    def group_norm(self, x0, x1, n_units, rows_per_unit, eps, gamma, beta, silu, ws, out, groups=32, prefetch=None):
        self._log("group_norm")
        stats = torch.empty(n_units, groups * 2)
        self.gn_stats(x0, x1, n_units, rows_per_unit, eps, ws, stats, groups)
        self.gn_apply(x0, x1, n_units, rows_per_unit, stats, gamma, beta, silu, out, groups)

    def group_norm_cs_ws_floats(self, n_units, rows_per_unit, groups):
        return 8

    def group_norm_cs(self, cs0, cs1, x0, x1, n_units, rows_per_unit, eps, gamma, beta, silu, ws, out, groups=32, prefetch=None):
        """GroupNorm(+SiLU) with the statistics taken from the producers' column statistics cs [rows / 32, C, 2]."""
        self._log("group_norm_cs")
        assert rows_per_unit % 32 == 0 and (x1 is None) == (cs1 is None)
        x = self._cat(x0, x1)
        C = x.shape[1]
        cs = cs0.float().view(-1, x0.shape[1], 2)
        if cs1 is not None:
            cs = torch.cat([cs, cs1.float().view(-1, x1.shape[1], 2)], dim=1)
        assert cs.shape[0] == n_units * rows_per_unit // 32
        sums = cs.view(n_units, rows_per_unit // 32, groups, C // groups, 2).sum(dim=(1, 3))    # [units, groups, 2]
        cnt = rows_per_unit * (C // groups)
        mean = sums[:, :, 0] / cnt
        var = (sums[:, :, 1] / cnt - mean * mean).clamp_min(0.0)
        stats = torch.stack([mean, 1.0 / torch.sqrt(var + eps)], dim=2).reshape(n_units, groups * 2)
        self.gn_apply(x0, x1, n_units, rows_per_unit, stats, gamma, beta, silu, out, groups)
        self.calls.pop()   # (one logical op)

    @staticmethod
    def gn_coef_cs_supported(cs0, cs1, c0, c1, n_units, rows_per_unit, groups=32):
        """Shape rules of t2v_gn_coef_cs_supported that do not depend on the thread layout (the emulation takes every such shape)."""
        cpg = (c0 + c1) // groups
        return rows_per_unit % 32 == 0 and (c0 + c1) % groups == 0 and cpg % 2 == 0 and c0 % 2 == 0

    def gn_coef_cs(self, cs0, cs1, c0, c1, n_units, rows_per_unit, eps, gamma, beta, coef, groups=32):
        """coef [n_units, 2, C]: rstd gamma / beta - mean rstd gamma per channel, from the producers' column statistics."""
        self._log("gn_coef_cs")
        C = c0 + c1
        cs = cs0.float().view(-1, c0, 2)
        if cs1 is not None:
            cs = torch.cat([cs, cs1.float().view(-1, c1, 2)], dim=1)
        sums = cs.view(n_units, rows_per_unit // 32, groups, C // groups, 2).sum(dim=(1, 3))
        cnt = rows_per_unit * (C // groups)
        mean = sums[:, :, 0] / cnt
        rstd = 1.0 / torch.sqrt((sums[:, :, 1] / cnt - mean * mean).clamp_min(0.0) + eps)
        a = rstd.repeat_interleave(C // groups, dim=1) * gamma.float()[None, :]
        b = beta.float()[None, :] - mean.repeat_interleave(C // groups, dim=1) * a
        coef.view(n_units, 2, C).copy_(torch.stack([a, b], dim=1))

    def gn_stats_cs(self, cs0, cs1, c0, c1, n_units, rows_per_unit, eps, ws, stats, groups=32):
        """(mean, rstd) per (unit, group) from the producers' column statistics."""
        self._log("gn_stats_cs")
        assert rows_per_unit % 32 == 0
        cs = cs0.float().view(-1, c0, 2)
        if cs1 is not None:
            cs = torch.cat([cs, cs1.float().view(-1, c1, 2)], dim=1)
        C = cs.shape[1]
        assert cs.shape[0] == n_units * rows_per_unit // 32
        sums = cs.view(n_units, rows_per_unit // 32, groups, C // groups, 2).sum(dim=(1, 3))
        cnt = rows_per_unit * (C // groups)
        mean = sums[:, :, 0] / cnt
        var = (sums[:, :, 1] / cnt - mean * mean).clamp_min(0.0)
        stats.copy_(torch.stack([mean, 1.0 / torch.sqrt(var + eps)], dim=2).reshape(stats.shape))

    # ------------------------------------------------------------------------------------ backward pieces
    def gn_bwd_ws_floats(self, n_units, rows_per_unit, groups=32):
        return 8

    def gn_bwd(self, x, n_units, rows_per_unit, stats, gamma, beta, silu, dy, resid, ws, dx, groups=32, x1=None):
        self._log("gn_bwd")
        x = self._cat(x, x1)
        C = x.shape[1]
        cpg = C // groups
        st = stats.reshape(n_units, groups, 2).float()
        mean = st[:, :, 0].repeat_interleave(cpg, dim=1)[:, None, :]
        rstd = st[:, :, 1].repeat_interleave(cpg, dim=1)[:, None, :]
        xh = (x.float().reshape(n_units, rows_per_unit, C) - mean) * rstd
        g = dy.float().reshape(n_units, rows_per_unit, C)
        if silu:
            u = xh * gamma.float() + beta.float()
            sig = torch.sigmoid(u)
            g = g * sig * (1 + u * (1 - sig))
        g = g * gamma.float()
        gg = g.reshape(n_units, rows_per_unit, groups, cpg)
        xg = xh.reshape(n_units, rows_per_unit, groups, cpg)
        m1 = gg.mean(dim=(1, 3), keepdim=True)
        m2 = (gg * xg).mean(dim=(1, 3), keepdim=True)
        d = (rstd.reshape(n_units, 1, groups, cpg) * (gg - m1 - xg * m2)).reshape(-1, C)
        if resid is not None:
            d = d + resid.float()
        dx.copy_(d.to(dx.dtype))

    def layernorm_bwd(self, x, gamma, eps, dy, resid, dx):
        """dx = d/dx [LayerNorm(x)] . dy (+ resid); the affine bias does not enter."""
        self._log("layernorm_bwd")
        xf = x.float()
        mean = xf.mean(dim=1, keepdim=True)
        rstd = 1.0 / torch.sqrt(xf.var(dim=1, unbiased=False, keepdim=True) + eps)
        xh = (xf - mean) * rstd
        g = dy.float() * gamma.float()
        d = rstd * (g - g.mean(dim=1, keepdim=True) - xh * (g * xh).mean(dim=1, keepdim=True))
        if resid is not None:
            d = d + resid.float()
        dx.copy_(d.to(dx.dtype))

    def geglu_fwd(self, h, out):
        """h: [M, 2*inner] in 64-column groups [32 value | 32 gate] (the packed GEGLU projection); out = value * gelu(gate)."""
        self._log("geglu_fwd")
        g = h.float().reshape(h.shape[0], -1, 2, 32)
        out.copy_((g[:, :, 0] * F.gelu(g[:, :, 1])).reshape(h.shape[0], -1).to(out.dtype))

    def geglu_bwd(self, h, dy, dh):
        self._log("geglu_bwd")
        M = h.shape[0]
        g = h.float().reshape(M, -1, 2, 32)
        v, gate = g[:, :, 0], g[:, :, 1]
        d = dy.float().reshape(M, -1, 32)
        cdf = 0.5 * (1.0 + torch.erf(gate * 0.7071067811865476))
        pdf = torch.exp(-0.5 * gate * gate) * 0.3989422804014327
        out = torch.stack([d * gate * cdf, d * v * (cdf + gate * pdf)], dim=2)
        dh.copy_(out.reshape(M, -1).to(dh.dtype))

    def attn_temporal_bwd(self, q, k, v, do, dprobs, dq, dk, dv, n_clips, frames, hw, heads, scale):
        """Backward of attn_temporal: (dq, dk, dv) from d(out) and, optionally, d(probs) [(b p head)][F][F] fp32."""
        self._log("attn_temporal_bwd")
        inner = heads * 64

        def seqs(t):
            return t.float().reshape(n_clips, frames, hw, heads, 64).permute(0, 2, 3, 1, 4).reshape(-1, frames, 64)

        def rows(t):
            return t.reshape(n_clips, hw, heads, frames, 64).permute(0, 3, 1, 2, 4).reshape(-1, inner)

        Q, Kk, V, dO = seqs(q), seqs(k), seqs(v), seqs(do)
        P = (Q @ Kk.transpose(1, 2) * scale).softmax(dim=2)
        dP = dO @ V.transpose(1, 2)
        if dprobs is not None:
            dP = dP + dprobs.float()
        dS = P * (dP - (P * dP).sum(dim=2, keepdim=True))
        dq.copy_(rows(dS @ Kk * scale).to(dq.dtype))
        dk.copy_(rows(dS.transpose(1, 2) @ Q * scale).to(dk.dtype))
        dv.copy_(rows(P.transpose(1, 2) @ dO).to(dv.dtype))

    def scatter2x(self, src, n_img, h, w, H, W, out):
        """Adjoint of the stride-2 sampling: out[n, 2y, 2x] = src[n, y, x], zero elsewhere (H in {2h-1, 2h}, W likewise)."""
        self._log("scatter2x")
        C = src.shape[1]
        z = torch.zeros(n_img, H, W, C)
        z[:, 0:2 * h:2, 0:2 * w:2] = src.float().reshape(n_img, h, w, C)[:, :(H + 1) // 2, :(W + 1) // 2]
        out.copy_(z.reshape(-1, C).to(out.dtype))

    def add(self, a, b, out):
        self._log("add")
        out.copy_((a.float() + b.float()).to(out.dtype))

    def softmax_bwd_rows(self, p, dp, rows, n, n_pad, ld):
        self._log("softmax_bwd_rows")
        pv = _strided(p, rows, n_pad, ld, 0).float()
        dv = _strided(dp, rows, n_pad, ld, 0)
        d = dv.float()
        out = torch.zeros(rows, n_pad)
        dot = (pv[:, :n] * d[:, :n]).sum(dim=1, keepdim=True)
        out[:, :n] = pv[:, :n] * (d[:, :n] - dot)
        dv.copy_(out.to(dp.dtype))

    def transpose(self, src, rows, cols, out, batch=1, in_stride=0, out_stride=0):
        self._log("transpose")
        for b in range(batch):
            a = _strided(src, rows, cols, src.stride(0), b * in_stride)
            _strided(out, cols, rows, out.stride(0), b * out_stride).copy_(a.t())

    def attn_spatial_bwd(self, q, k, v, v_img_stride, v_head_stride, kt, qt, dot, dout, o, l2, dsum, dq, dk, dv, n_img, seq, heads, scale):
        """(dq, dk, dv) of softmax(scale q k^T) v per (image, head); the transposed operands / workspaces are the kernel's business."""
        self._log("attn_spatial_bwd")
        for img in range(n_img):
            r = slice(img * seq, (img + 1) * seq)
            for hd in range(heads):
                c = slice(hd * 64, (hd + 1) * 64)
                Q, Kk, dO = q[r, c].float(), k[r, c].float(), dout[r, c].float()
                V = torch.as_strided(v, (seq, 64), (v.stride(0), 1), v.storage_offset() + img * v_img_stride + hd * v_head_stride).float()
                P = (Q @ Kk.t() * scale).softmax(dim=1)
                dP = dO @ V.t()
                dS = P * (dP - (P * dP).sum(dim=1, keepdim=True))
                dq[r, c] = (dS @ Kk * scale).to(dq.dtype)
                dk[r, c] = (dS.t() @ Q * scale).to(dk.dtype)
                dv[r, c] = (P.t() @ dO).to(dv.dtype)

    # ---- base-weight gradients for full fine-tuning (csrc/full_grad.hip) -----------------------------------------------------------
    @staticmethod
    def im2col_rows(mode, n_img, h, w):
        if mode in (nt.GEMM_CONV3X3, nt.GEMM_TCONV3):
            return n_img * h * w
        if mode == nt.GEMM_CONV3X3_S2:
            return n_img * ((h - 1) // 2 + 1) * ((w - 1) // 2 + 1)
        if mode == nt.GEMM_CONV3X3_S2_PAD01:
            return n_img * ((h - 2) // 2 + 1) * ((w - 2) // 2 + 1)
        if mode == nt.GEMM_CONV3X3_UP2:
            return n_img * 4 * h * w
        return -1

    def repack_conv(self, w, out, kind):
        """t2v_repack_conv_f32: kind 0 out[n][t*C + c] = w[n][c][t]; kind 1 out[c][(taps-1-t)*N + n] = w[n][c][t]."""
        self._log("repack_conv")
        N, Cc = w.shape[0], w.shape[1]
        w3 = w.detach().reshape(N, Cc, -1)
        if kind == 0:
            out.copy_(w3.permute(0, 2, 1).reshape(N, -1))
        else:
            out.copy_(w3.flip(2).permute(1, 2, 0).reshape(Cc, -1))

    def im2col(self, x0, x1, mode, n_img, h, w, frames, out):
        """out[m][tap * C + c] = x[src(m, tap)][c]: F.unfold of the (padded / upsampled) image, re-ordered tap-major."""
        self._log("im2col")
        x = self._cat(x0, x1).float()
        C = x.shape[1]
        if mode == nt.GEMM_TCONV3:
            b = n_img // frames
            x5 = F.pad(x.reshape(b, frames, h * w, C), (0, 0, 0, 0, 1, 1))            # frames padded by one on both sides
            cols = torch.stack([x5[:, t:t + frames] for t in range(3)], dim=3)        # b f p tap c
            out[:, :3 * C].copy_(cols.reshape(-1, 3 * C).to(out.dtype))
            return
        x4 = x.reshape(n_img, h, w, C).permute(0, 3, 1, 2)
        if mode == nt.GEMM_CONV3X3:
            u = F.unfold(x4, 3, padding=1)
        elif mode == nt.GEMM_CONV3X3_S2:
            u = F.unfold(x4, 3, padding=1, stride=2)
        elif mode == nt.GEMM_CONV3X3_S2_PAD01:
            u = F.unfold(F.pad(x4, (0, 1, 0, 1)), 3, stride=2)
        elif mode == nt.GEMM_CONV3X3_UP2:
            u = F.unfold(F.interpolate(x4, scale_factor=2, mode="nearest"), 3, padding=1)
        else:
            raise ValueError(mode)
        # unfold: [n, C * 9, L] with the channel the slow index -> [n * L, 9, C]
        u = u.reshape(n_img, C, 9, -1).permute(0, 3, 2, 1).reshape(-1, 9 * C)
        out[:, :9 * C].copy_(u.to(out.dtype))

    def norm_affine_grad_ws_floats(self, rows, sum_rows, channels):
        return 8

    def norm_affine_grad(self, x0, x1, dy, *, kind, sum_rows, ws, dgamma=None, dbeta=None, rows_per_unit=0, groups=0, stats=None, eps=0.0,
                         gamma=None, beta=None, silu=False):
        self._log("norm_affine_grad")
        rows = dy.shape[0]
        g = dy.float()
        C = g.shape[1]
        xh = None
        if kind == 0:
            x = self._cat(x0, x1).float()
            units, cpg = rows // rows_per_unit, C // groups
            st = stats.reshape(units, groups, 2).float()
            mean = st[:, :, 0].repeat_interleave(cpg, dim=1)[:, None, :]
            rstd = st[:, :, 1].repeat_interleave(cpg, dim=1)[:, None, :]
            xh = ((x.reshape(units, rows_per_unit, C) - mean) * rstd).reshape(rows, C)
        elif kind == 1:
            x = self._cat(x0, x1).float()
            xh = (x - x.mean(dim=1, keepdim=True)) / torch.sqrt(x.var(dim=1, unbiased=False, keepdim=True) + eps)
        if silu and kind != 2:
            z = xh * gamma.float() + beta.float()
            sig = torch.sigmoid(z)
            g = g * sig * (1 + z * (1 - sig))
        n_out = rows // sum_rows
        if dgamma is not None:
            dgamma.copy_((g * xh).reshape(n_out, sum_rows, C).sum(dim=1) if xh is not None else torch.zeros(n_out, C))
        if dbeta is not None:
            dbeta.copy_(g.reshape(n_out, sum_rows, C).sum(dim=1))

    def wgrad_tn(self, a, b, out, alpha=1.0, splits=0):
        self._log("wgrad_tn")
        if self.strict:
            assert a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0 and a.storage_offset() % 8 == 0 and b.storage_offset() % 8 == 0
        out.copy_((a.float().t() @ b.float()) * alpha)

    def wgrad_tn_group(self, problems):
        """[(a, b, out, alpha)]: the weight gradients of one LoRA group as one op (t2v_wgrad_tn_group)."""
        self._log("wgrad_tn_group")
        for a, b, out, alpha in problems:
            if self.strict:
                assert a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0 and a.storage_offset() % 8 == 0 and b.storage_offset() % 8 == 0
                assert out.dtype == torch.float32 and out.shape == (a.shape[1], b.shape[1])
            out.copy_((a.float().t() @ b.float()) * alpha)

    def transpose_pad(self, src, rows, cols, out, batch=1, in_stride=0, out_stride=0):
        self._log("transpose_pad")
        rp = (rows + 63) // 64 * 64
        assert cols % 8 == 0 and src.stride(0) % 8 == 0 and out.stride(0) % 8 == 0 and out.stride(0) >= rp
        for b in range(batch):
            a = _strided(src, rows, cols, src.stride(0), b * in_stride)
            o = _strided(out, cols, rp, out.stride(0), b * out_stride)
            o[:, :rows] = a.t()
            o[:, rows:] = 0

    def sumpool2x2(self, src, n_img, h, w, out):
        self._log("sumpool2x2")
        C = src.shape[1]
        v = src.float().reshape(n_img, h, 2, w, 2, C).sum(dim=(2, 4))
        out.copy_(v.reshape(-1, C).to(out.dtype))

    def layernorm(self, x, gamma, beta, eps, out):
        self._log("layernorm")
        out.copy_(F.layer_norm(x.float(), (x.shape[1],), gamma.float(), beta.float(), eps).to(out.dtype))

    def softmax_rows(self, s, rows, n, n_pad, ld):
        self._log("softmax_rows")
        v = _strided(s, rows, n_pad, ld, 0)
        p = torch.zeros(rows, n_pad)
        p[:, :n] = v[:, :n].float().softmax(dim=1)
        v.copy_(p.to(s.dtype))

    # ------------------------------------------------------------------------------------ attention
    def attn_spatial(self, q, k, vt, ld_vt, out, n_img, seq_q, seq_kv, heads, kv_div, scale, vt_img_stride=0):
        self._log("attn_spatial")
        assert ld_vt >= ((seq_kv + 63) // 64) * 64
        vt_img_stride = vt_img_stride or heads * 64 * ld_vt
        for img in range(n_img):
            ikv = img // kv_div
            for hd in range(heads):
                Q = q[img * seq_q:(img + 1) * seq_q, hd * 64:(hd + 1) * 64].float()
                Kk = k[ikv * seq_kv:(ikv + 1) * seq_kv, hd * 64:(hd + 1) * 64].float()
                Vt = _strided(vt, 64, seq_kv, ld_vt, ikv * vt_img_stride + hd * 64 * ld_vt).float()
                P = (Q @ Kk.t() * scale).softmax(dim=1)
                out[img * seq_q:(img + 1) * seq_q, hd * 64:(hd + 1) * 64] = (P @ Vt.t()).to(out.dtype)

    def attn_temporal(self, q, k, v, out, n_clips, frames, hw, heads, scale, probs=None):
        self._log("attn_temporal")
        inner = heads * 64

        def seqs(t):  # rows ((b f) p) -> (b p head) f d
            return t.float().reshape(n_clips, frames, hw, heads, 64).permute(0, 2, 3, 1, 4).reshape(-1, frames, 64)

        Q, Kk, V = seqs(q), seqs(k), seqs(v)
        P = (Q @ Kk.transpose(1, 2) * scale).softmax(dim=2)
        if probs is not None:
            probs.copy_(P)
        O = (P @ V).reshape(n_clips, hw, heads, frames, 64).permute(0, 3, 1, 2, 4).reshape(-1, inner)
        out.copy_(O.to(out.dtype))

    # ------------------------------------------------------------------------------------ layout / elementwise
    def ncfhw_to_tokens(self, x, out):
        self._log("ncfhw_to_tokens")
        b, c, f, h, w = x.shape
        out[:, :c] = x.float().permute(0, 2, 3, 4, 1).reshape(-1, c).to(out.dtype)

    def tokens_to_ncfhw(self, tok, out):
        self._log("tokens_to_ncfhw")
        b, c, f, h, w = out.shape
        out.copy_(tok[:, :c].float().reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3).to(out.dtype))

    def timestep_embedding(self, t, dim, guidance_style, out):
        self._log("timestep_embedding")
        half = dim // 2
        tv = t.float()
        if guidance_style:
            freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000.0) / (half - 1)))
            a = tv[:, None] * 1000.0 * freq[None]
            e = torch.cat([torch.sin(a), torch.cos(a)], dim=1)
        else:
            freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
            a = tv[:, None] * freq[None]
            e = torch.cat([torch.cos(a), torch.sin(a)], dim=1)
        out[:, :2 * half] = e.to(out.dtype)

    def silu(self, x, out):
        self._log("silu")
        out.copy_(F.silu(x.float()).to(out.dtype))

    def cast(self, x, out):
        self._log("cast")
        out.copy_(x.reshape(out.shape).to(out.dtype))

    def fill_zero(self, t):
        self._log("fill_zero")
        t.zero_()

    def lincomb3(self, x, y, z, ca, cb, cc, out):
        self._log("lincomb3")
        nb = len(ca)
        shp = (nb,) + (1,) * (x.dim() - 1)
        v = torch.tensor(ca).reshape(shp) * x
        if y is not None:
            v = v + torch.tensor(cb).reshape(shp) * y
        if z is not None:
            v = v + torch.tensor(cc).reshape(shp) * z
        out.copy_(v)

    def gather(self, src, idx, out, alpha=1.0, accumulate=False):
        self._log("gather")
        assert src.dtype == torch.float32 and idx.dtype == torch.int32 and idx.numel() == out.numel()
        assert out.is_contiguous() and src.is_contiguous()
        j = idx.long()
        v = torch.where(j >= 0, src.reshape(-1)[j.clamp_min(0)] * alpha, torch.zeros((), dtype=torch.float32))
        o = out.view(-1)
        if accumulate:
            o.copy_(torch.where(j >= 0, o.float() + v, o.float()).to(out.dtype))
        else:
            o.copy_(v.to(out.dtype))

    @staticmethod
    def dropout_keep(seed, site, rows, ncols, p):
        """The device kernel's mask, bit for bit (csrc/common.h): one splitmix64 word per quad of adjacent elements of the row-major
        [rows][ncols] matrix, 16 bits per element against (p * 2^32) >> 16."""
        import numpy as np
        assert ncols % 2 == 0
        m64 = np.uint64
        n = rows * ncols
        with np.errstate(over="ignore"):
            quad = np.arange((n + 3) // 4, dtype=np.uint64)
            z = m64(seed & 0xFFFFFFFFFFFFFFFF) + m64(site) * m64(0x9E3779B97F4A7C15) + quad * m64(0xD1B54A32D192ED03)
            z ^= z >> m64(30)
            z *= m64(0xBF58476D1CE4E5B9)
            z ^= z >> m64(27)
            z *= m64(0x94D049BB133111EB)
            z ^= z >> m64(31)
        t = p * 4294967296.0
        thr = np.uint64((0xFFFFFFFF if t >= 4294967295.0 else int(t)) >> 16)
        bits = np.stack([(z >> m64(16 * e)) & m64(0xFFFF) for e in range(4)], axis=1).reshape(-1)[:n]
        return torch.from_numpy((bits >= thr).reshape(rows, ncols))

    def dropout(self, x, resid, out, ncols, p, seed, site):
        self._log("dropout")
        keep = self.dropout_keep(int(seed.reshape(-1)[0]), site, x.shape[0], ncols, p)
        if getattr(self, "masks", None) is not None:
            self.masks[site] = keep
        v = torch.where(keep, x[:, :ncols].float() * nt.dropout_inv_keep(p), torch.zeros(()))
        if resid is not None:
            v = v + resid[:, :ncols].float()
        out[:, :ncols] = v.to(out.dtype)

    def lcm_step(self, x, eps, noise, sa_t, sb_t, c_skip, c_out, sa_p, sb_p, prev, denoised):
        self._log("lcm_step")
        x0 = (x - sb_t * eps.float()) / sa_t
        den = c_out * x0 + c_skip * x
        denoised.copy_(den)
        prev.copy_(den if noise is None else sa_p * den + sb_p * noise)


class ReplayOps:
    """TEST ONLY: the emulated ops behind the NATIVE backend's record / replay protocol (``is_native``, ``recording`` lists,
    ``replay``): the engines then take the code path they take on the GPU — launches recorded once with their operand views
    baked in, later calls only refresh the static input buffers and re-issue the recorded list — instead of re-running their
    Python closures every time.  Catches what only that mode can get wrong: state that lives in Python during recording but
    not during a replay, buffers recycled between the forward and backward lists, inputs that are not static."""
    is_native = True
    _PURE = ("gn_ws_floats", "gn_bwd_ws_floats", "group_norm_ws_floats", "group_norm_cs_ws_floats", "gemm_fuse_supported", "ffn_fused_supported", "dropout_keep", "masks", "calls", "strict")

    def __init__(self, inner=None):
        self.inner = inner or EmuOps(strict=True)
        self.act_dtype = self.inner.act_dtype
        self.recording = None
        self.replays = 0

    def init(self):
        pass

    @staticmethod
    def stream():
        return None

    def replay(self, recording, stream, cache=True):
        self.replays += 1
        for fn, a, k in recording:
            fn(*a, **k)

    def record_host_call(self, fn, args, name):
        if self.recording is not None:
            self.recording.append((fn, args, {}))

    def __getattr__(self, name):
        fn = getattr(self.inner, name)
        if name in self._PURE or not callable(fn):
            return fn

        def call(*a, **k):
            if self.recording is not None:
                self.recording.append((fn, a, k))
            return fn(*a, **k)
        return call
