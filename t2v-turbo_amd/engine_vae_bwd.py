"""Native data gradient of the KL-VAE decode: d(loss)/d(latents) for the reward branch of the
v1 trainer (``train_t2v_turbo_v1_lora.py:1047-1098``: ``vae.decode(selected_latents)`` -> reward model -> backward into
the student's ``model_pred``; the VAE itself is frozen, so only dX is needed, never dW).

The forward pass is the decode engine's, recorded once together with a *tape*: every block returns a closure that
records its own backward launches.  What the tape keeps alive are the GroupNorm inputs and their (mean, rstd), and the
attention block's q / k / V^T / probabilities — everything else is recomputed from the incoming gradient:
  conv3x3 / 1x1 data gradients  = the same implicit-GEMM kernel on re-packed weights (flipped taps, swapped channels),
  nearest-x2 upsample + conv    = data-gradient conv at the high resolution, then a 2x2 sum-pool,
  GroupNorm(+swish)             = ``t2v_gn_bwd`` (slab partial sums of g and g*xhat, finish, apply; residual add fused),
  attention                     = five batched GEMMs around ``t2v_softmax_bwd_rows`` (bf16 transposes feed the GEMMs whose
                                  contraction index is the row index of the stored operands).
Forward and backward are two replayable launch lists (hipGraph-capturable) sharing one buffer pool."""
import torch


from . import native as nt
from .native import on_tensor_device
from .engine import Act, leaf_out_channels
from .engine_vae import VAEDecodeEngine
from .vae import AttnBlock


class VAEDecodeGradEngine(VAEDecodeEngine):
    @on_tensor_device
    def decode_frames_tape(self, z, scale):
        """Forward like ``decode_frames`` but keeps what ``backward`` needs; returns the video (b, 3, t, 8h, 8w)."""
        assert z.dim() == 5
        self._check_weights(self.vae)
        key = ("grad", tuple(z.shape), z.dtype, float(scale), z.device)
        plan = self.plans.get(key)
        if plan is None:
            plan = self._own(self._record_grad(z, scale))
            self._keep_plan(key, plan)
            if getattr(self.ops, "is_native", False):
                # recording executed the backward list once (on a zero gradient) and that recycled the saved forward
                # buffers: run the forward list again so the tape holds this call's activations
                self._replay(plan, "rec")
        else:
            plan["static"]["z"].copy_(z)
            self._replay(plan, "rec")
        plan["fwd_id"] = plan.get("fwd_id", 0) + 1
        self._last = plan
        return plan["out"].clone()

    @on_tensor_device
    def backward(self, dout):
        """d(loss)/dz for the most recent ``decode_frames_tape`` call."""
        plan = self._last
        if plan.get("bwd_id") == plan["fwd_id"]:
            raise RuntimeError("VAE decode gradient: backward was already run for this forward (its saved activations are gone)")
        plan["bwd_id"] = plan["fwd_id"]
        plan["static"]["dout"].copy_(dout)
        self._replay(plan, "rec_bwd")
        return plan["dz"].clone()

    def _replay(self, plan, which):
        ops = self.ops
        if getattr(ops, "is_native", False):
            ops.replay(plan[which], ops.stream())
        else:
            plan["fn" if which == "rec" else "fn_bwd"]()

    # ---- recording ----------------------------------------------------------------------------------------
    def _record_grad(self, z, scale):
        ops = self.ops
        self._begin(z.device)
        dec = self.vae.decoder
        b, zc, t, h, w = z.shape
        up = 2 ** (dec.num_resolutions - 1)
        cout = leaf_out_channels(dec.conv_out)
        st = {"z": z.detach().clone().contiguous(),
              "dout": torch.zeros(b, cout, t, h * up, w * up, dtype=z.dtype, device=z.device)}
        plan = {"static": st, "out": torch.empty_like(st["dout"]), "dz": torch.empty_like(st["z"]), "runs": 0}
        native = getattr(ops, "is_native", False)
        self.tape = []

        def fwd():
            self.tape.clear()
            self._forward_tape(st["z"], scale, plan["out"])

        def bwd():
            self._backward_tape(st["dout"], plan["dz"])

        if native:
            ops.init()
            ops.recording = []
            try:
                fwd()
            finally:
                plan["rec"] = ops.recording
                ops.recording = None
            ops.recording = []
            try:
                bwd()
            finally:
                plan["rec_bwd"] = ops.recording
                ops.recording = None
        else:  # emulation backend (tests): the closures themselves are the plan; every forward is followed by one backward
            fwd()
            plan["fn"], plan["fn_bwd"] = fwd, bwd
        plan["pool_bytes"] = self.pool.bytes
        return plan

    # ---- helpers ------------------------------------------------------------------------------------------------
    def gn_t(self, x, norm, units, rows, silu):
        """GroupNorm(+swish) keeping (mean, rstd); x stays alive for the backward."""
        ops = self.ops
        G = norm.num_groups
        ws = self.buf(1, max(ops.gn_ws_floats(units, rows, G), 1), torch.float32)
        stats = self.buf(units, G * 2, torch.float32)
        ops.gn_stats(x, None, units, rows, norm.eps, ws, stats, G)
        out = self.buf(x.shape[0], x.shape[1])
        ops.gn_apply(x, None, units, rows, stats, self.pk.f32(norm.weight), self.pk.f32(norm.bias), silu, out, G)
        self.pool.put(ws)
        return out, stats

    def gn_b(self, x, norm, units, rows, silu, stats, dy, resid=None):
        ops = self.ops
        G = norm.num_groups
        ws = self.buf(1, max(ops.gn_bwd_ws_floats(units, rows, G), 1), torch.float32)
        dx = self.buf(x.shape[0], x.shape[1])
        ops.gn_bwd(x, units, rows, stats, self.pk.f32(norm.weight), self.pk.f32(norm.bias), silu, dy, resid, ws, dx, G)
        self.pool.put(ws)
        return dx

    # ---- forward with tape ------------------------------------------------------------------------------------
    def _forward_tape(self, z, scale, out):
        ops, pk, vae = self.ops, self.pk, self.vae
        dec = vae.decoder
        b, zc, t, h, w = z.shape
        n_img = b * t
        assert zc == 4, "the gradient path is built for the 4-channel KL-f8 latent"
        zt = self.buf(n_img * h * w, zc)
        ops.ncfhw_to_tokens(z, zt)
        pq = vae.post_quant_conv

        def pq_weights():
            wq = pq.weight.detach().float().reshape(pq.weight.shape[0], zc) * scale
            w3 = torch.zeros(8, 9, zc, dtype=torch.float32)
            w3[: wq.shape[0], 4, :] = wq.cpu()
            b8 = torch.zeros(8, dtype=torch.float32)
            b8[: wq.shape[0]] = pq.bias.detach().float().cpu()
            return w3.reshape(8, -1).to(self.device).contiguous(), b8.to(self.device)

        def pq_dgrad_weights():  # d(zt)[ci] = sum_co d(z8)[co] * wq[co][ci] * scale, as the centre tap of a 4 -> 8 direct conv
            wq = pq.weight.detach().float().reshape(pq.weight.shape[0], zc) * scale
            w3 = torch.zeros(8, 9, 4, dtype=torch.float32)
            w3[:zc, 4, : wq.shape[0]] = wq.t().cpu()
            return w3.reshape(8, -1).to(self.device).contiguous()

        w3, b8 = pk._memo(("pq", id(pq), float(scale)), pq_weights)
        z8 = self.buf(n_img * h * w, 8)
        ops.conv_small(zt, n_img, h, w, w3, b8, z8)
        h0 = self.buf(n_img * h * w, leaf_out_channels(dec.conv_in))
        ops.conv_small(z8, n_img, h, w, pk.small_conv(dec.conv_in, cin_pad=8), pk.bias(dec.conv_in), h0)
        self.pool.put(zt, z8)

        def entry_bwd(dy, dz_out):
            # conv_in data gradient (512 -> the 4 real post-quant channels), then post_quant_conv^T, then back to NCFHW
            wd = pk._memo(("conv_in_dgrad", id(dec.conv_in)), lambda: pk.conv_dgrad(dec.conv_in)[:4].contiguous())
            d4 = self.conv(dy, dec.conv_in, nt.GEMM_CONV3X3, w=wd, bias=None)
            self.pool.put(dy.t)
            d8 = self.buf(n_img * h * w, 8)
            ops.conv_small(d4.t, n_img, h, w, pk._memo(("pq_dgrad", id(pq), float(scale)), pq_dgrad_weights), None, d8)
            self.pool.put(d4.t)
            ops.tokens_to_ncfhw(d8, dz_out)
            self.pool.put(d8)

        x = Act(h0, n_img, h, w)
        x = self.resnet_block_t(dec.mid.block_1, x)
        if isinstance(dec.mid.attn_1, AttnBlock):
            x = self.attn_block_t(dec.mid.attn_1, x)
        x = self.resnet_block_t(dec.mid.block_2, x)
        for lvl in reversed(range(dec.num_resolutions)):
            for ib in range(dec.num_res_blocks + 1):
                x = self.resnet_block_t(dec.up[lvl].block[ib], x)
                assert len(dec.up[lvl].attn) == 0, "attention inside the up path is not used by the KL-f8 config"
            if lvl != 0:
                x = self.upsample_t(dec.up[lvl].upsample, x)
        xin = x
        tt, st_out = self.gn_t(x.t, dec.norm_out, n_img, x.h * x.w, True)
        y = self.conv(Act(tt, n_img, x.h, x.w), dec.conv_out, nt.GEMM_CONV3X3, out_dtype=torch.float32)
        self.pool.put(tt)
        ops.tokens_to_ncfhw(y.t, out)
        self.pool.put(y.t)
        H, W = x.h, x.w

        def exit_bwd(dout):
            d4 = self.buf(n_img * H * W, 4)
            ops.fill_zero(d4)
            ops.ncfhw_to_tokens(dout, d4)  # 3 image channels into a zero-padded 4-channel row
            dt = self.buf(n_img * H * W, xin.C)
            ops.conv_small(d4, n_img, H, W, pk.small_conv_dgrad(dec.conv_out, cin_pad=4), None, dt)
            self.pool.put(d4)
            dx = self.gn_b(xin.t, dec.norm_out, n_img, H * W, True, st_out, dt)
            self.pool.put(dt, xin.t, st_out)
            return Act(dx, n_img, H, W)

        self.entry_bwd, self.exit_bwd = entry_bwd, exit_bwd

    def _backward_tape(self, dout, dz_out):
        d = self.exit_bwd(dout)
        for fn in reversed(self.tape):
            d = fn(d)
        self.entry_bwd(d, dz_out)

    def resnet_block_t(self, rb, x):
        hw, n = x.h * x.w, x.n_img
        t1, st1 = self.gn_t(x.t, rb.norm1, n, hw, True)
        h1 = self.conv(Act(t1, n, x.h, x.w), rb.conv1, nt.GEMM_CONV3X3)
        self.pool.put(t1)
        t2, st2 = self.gn_t(h1.t, rb.norm2, n, hw, True)
        skip, own = x.t, False
        shortcut = rb.in_channels != rb.out_channels
        if shortcut:
            assert not rb.use_conv_shortcut, "3x3 conv shortcuts are not used by the KL-f8 config"
            skip = self.linear(x.t, rb.nin_shortcut)
            own = True
        h2 = self.conv(Act(t2, n, x.h, x.w), rb.conv2, nt.GEMM_CONV3X3, residual=skip)
        self.pool.put(t2)
        if own:
            self.pool.put(skip)

        def bwd(dy):
            d_t2 = self.conv(dy, rb.conv2, nt.GEMM_CONV3X3, w=self.pk.conv_dgrad(rb.conv2), bias=None)
            d_h1 = self.gn_b(h1.t, rb.norm2, n, hw, True, st2, d_t2.t)
            self.pool.put(d_t2.t, h1.t, st2)
            d_t1 = self.conv(Act(d_h1, n, x.h, x.w), rb.conv1, nt.GEMM_CONV3X3, w=self.pk.conv_dgrad(rb.conv1), bias=None)
            self.pool.put(d_h1)
            if shortcut:
                d_skip = self.linear(dy.t, None, w=self.pk.mat_t(rb.nin_shortcut), bias=None)
            else:
                d_skip = dy.t
            dx = self.gn_b(x.t, rb.norm1, n, hw, True, st1, d_t1.t, resid=d_skip)
            self.pool.put(d_t1.t, x.t, st1, dy.t)
            if shortcut:
                self.pool.put(d_skip)
            return Act(dx, n, x.h, x.w)

        self.tape.append(bwd)
        return h2

    def upsample_t(self, ups, x):
        assert ups.with_conv
        y = self.conv(x, ups.conv, nt.GEMM_CONV3X3_UP2)
        self.pool.put(x.t)
        n, h, w, cin = x.n_img, x.h, x.w, x.C

        def bwd(dy):
            d_up = self.conv(dy, ups.conv, nt.GEMM_CONV3X3, w=self.pk.conv_dgrad(ups.conv), bias=None)  # at 2h x 2w
            self.pool.put(dy.t)
            dx = self.buf(n * h * w, cin)
            self.ops.sumpool2x2(d_up.t, n, h, w, dx)
            self.pool.put(d_up.t)
            return Act(dx, n, h, w)

        self.tape.append(bwd)
        return y

    def attn_block_t(self, ab, x):
        ops, pk = self.ops, self.pk
        c, seq, n = x.C, x.h * x.w, x.n_img
        kp = ((seq + 63) // 64) * 64
        alpha = float(int(c) ** -0.5)
        t, st = self.gn_t(x.t, ab.norm, n, seq, False)
        q = self.linear(t, ab.q)
        k = self.linear(t, ab.k)
        vt = self.buf(n * c, kp)
        if kp != seq:
            ops.fill_zero(vt)
        ops.gemm(pk.mat(ab.v), t, vt, M=c, N=seq, batch=n, w_strides=(seq * t.stride(0), 0), o_strides=(c * kp, 0))
        s = self.buf(n * seq, kp)
        ops.gemm(q, k, s, M=seq, N=seq, alpha=alpha, batch=n, a_strides=(seq * q.stride(0), 0),
                 w_strides=(seq * k.stride(0), 0), o_strides=(seq * kp, 0))
        ops.softmax_rows(s, n * seq, seq, kp, kp)
        o = self.buf(n * seq, c)
        ops.gemm(s, vt, o, M=seq, N=c, batch=n, a_strides=(seq * kp, 0), w_strides=(c * kp, 0), o_strides=(seq * c, 0),
                 bias=pk.bias(ab.v))
        out = self.linear(o, ab.proj_out, residual=x.t)
        self.pool.put(t, o)

        def tposed(src, rows, cols, in_stride):
            """[n][rows][cols] -> [n][cols (padded to kp when it is a key/query index)][kp], zero where never written."""
            out_rows = kp if cols == kp else cols
            dst = self.buf(n * out_rows, kp)
            if kp != seq:
                ops.fill_zero(dst)
            ops.transpose(src, rows, cols, dst, batch=n, in_stride=in_stride, out_stride=out_rows * kp)
            return dst

        def bwd(dy):  # dy [n*seq, c]; every GEMM below contracts over a K-contiguous index (hence the transposes)
            d_o = self.linear(dy.t, None, w=pk.mat_t(ab.proj_out), bias=None)
            # dP[q][kv] = sum_c d_o[q][c] V[kv][c]
            v_tok = self.buf(n * kp, c)
            ops.transpose(vt, c, kp, v_tok, batch=n, in_stride=c * kp, out_stride=kp * c)
            dp = self.buf(n * seq, kp)
            ops.gemm(d_o, v_tok, dp, M=seq, N=kp, batch=n, a_strides=(seq * c, 0), w_strides=(kp * c, 0), o_strides=(seq * kp, 0))
            self.pool.put(v_tok, vt)
            # dV[kv][c] = sum_q P[q][kv] d_o[q][c]: A = P^T [kv][q], W = d_o^T [c][q]
            pT = tposed(s, seq, kp, seq * kp)
            doT = tposed(d_o, seq, c, seq * c)
            self.pool.put(d_o)
            d_v = self.buf(n * seq, c)
            ops.gemm(pT, doT, d_v, M=seq, N=c, batch=n, a_strides=(kp * kp, 0), w_strides=(c * kp, 0), o_strides=(seq * c, 0))
            self.pool.put(pT, doT)
            # dS = P * (dP - rowsum(dP * P)), in place on dP
            ops.softmax_bwd_rows(s, dp, n * seq, seq, kp, kp)
            self.pool.put(s)
            # dQ[q][c] = alpha * sum_kv dS[q][kv] K[kv][c]: A = dS, W = K^T [c][kv]
            kT = tposed(k, seq, c, seq * c)
            d_q = self.buf(n * seq, c)
            ops.gemm(dp, kT, d_q, M=seq, N=c, alpha=alpha, batch=n, a_strides=(seq * kp, 0), w_strides=(c * kp, 0),
                     o_strides=(seq * c, 0))
            self.pool.put(kT, k)
            # dK[kv][c] = alpha * sum_q dS[q][kv] Q[q][c]: A = dS^T [kv][q], W = Q^T [c][q]
            dsT = tposed(dp, seq, kp, seq * kp)
            qT = tposed(q, seq, c, seq * c)
            self.pool.put(dp, q)
            d_k = self.buf(n * seq, c)
            ops.gemm(dsT, qT, d_k, M=seq, N=c, alpha=alpha, batch=n, a_strides=(kp * kp, 0), w_strides=(c * kp, 0),
                     o_strides=(seq * c, 0))
            self.pool.put(dsT, qT)
            # back through the three 1x1 projections into d(t), then GroupNorm (no activation) + the residual path
            d_t = self.linear(d_q, None, w=pk.mat_t(ab.q), bias=None)
            d_t2 = self.linear(d_k, None, w=pk.mat_t(ab.k), bias=None, residual=d_t)
            d_t3 = self.linear(d_v, None, w=pk.mat_t(ab.v), bias=None, residual=d_t2)
            self.pool.put(d_q, d_k, d_v, d_t, d_t2)
            dx = self.gn_b(x.t, ab.norm, n, seq, False, st, d_t3, resid=dy.t)
            self.pool.put(d_t3, x.t, st, dy.t)
            return Act(dx, n, x.h, x.w)

        self.tape.append(bwd)
        return Act(out, n, x.h, x.w)

