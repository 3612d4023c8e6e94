"""Native (HIP) execution of the ModelScope denoiser ``ms_unet3d.UNet3DConditionModel``.

Same record-once / replay (hipGraph) engine and the same kernels as the VideoCrafter2 path: the transformer blocks,
temporal conv layers and attention kernels are inherited from ``engine.UNetEngine`` unchanged; only the wiring
(diffusers block containers, ``ResnetBlock2D``, ``TimestepEmbedding.cond_proj``) is specific.  Every ``time_emb_proj``
Linear of the network is one stacked GEMM, the text K / V^T projections of all cross-attention layers two."""
import torch
import torch.nn as nn

from . import native as nt
from .engine import Act, UNetEngine, leaf_out_channels
from .ms_unet3d import ResnetBlock2D


class MSUNetEngine(UNetEngine):
    def _forward(self, st, out):
        m, ops, pk = self.model, self.ops, self.pk
        B, F = self.B, self.F
        x = st["x"]
        _, Cin, _, H, W = x.shape
        mc = m.model_channels
        L, D = st["ctx"].shape[1], st["ctx"].shape[2]
        te = m.time_embedding
        # ---- time embedding (unet_3d_condition.py:396-408): cos||sin -> (+ cond_proj(w)) -> linear_1 -> SiLU -> linear_2
        t_emb = self.buf(B, mc)
        ops.timestep_embedding(st["ts"], mc, False, t_emb)
        emb_in = t_emb
        if "tc" in st:
            if te.cond_proj is None:
                raise ValueError("timestep_cond given but the model was built without time_cond_proj_dim")
            tcb = self.buf(B, st["tc"].shape[1])
            ops.cast(st["tc"], tcb)
            emb_in = self.linear(tcb, te.cond_proj, residual=t_emb)
        e1 = self.linear(emb_in, te.linear_1, act=nt.ACT_SILU)
        emb = self.linear(e1, te.linear_2)
        emb_s = self.buf(B, emb.shape[1])
        ops.silu(emb, emb_s)
        resnets = [mod for mod in m.modules() if isinstance(mod, ResnetBlock2D)]
        self.emb_off, off = {}, 0
        for rb in resnets:
            self.emb_off[id(rb)] = off
            off += rb.out_channels
        lins = [rb.time_emb_proj for rb in resnets]
        w_all = pk.cat_mats(lins, "ms_emb_all")
        b_all = pk._memo(("ms_emb_all_bias",) + tuple(id(l) for l in lins),
                         lambda: torch.cat([pk.bias(l) for l in lins]).contiguous())
        self.emb_all = self.linear(emb_s, None, w=w_all, bias=b_all, out_dtype=torch.float32)
        # ---- text context, shared by all frames of a clip
        self.ctx = self.buf(B * L, D)
        ops.cast(st["ctx"], self.ctx)
        self.ctx_len = L
        self.ctx_kv = {}
        # ---- conv_in on the 4-channel latent, then the input temporal transformer
        xt = self.buf(B * F * H * W, Cin)
        ops.ncfhw_to_tokens(x, xt)
        h0 = self.buf(B * F * H * W, leaf_out_channels(m.conv_in))
        ops.conv_small(xt, B * F, H, W, pk.small_conv(m.conv_in), pk.bias(m.conv_in), h0)
        h = Act(h0, B * F, H, W)
        nh = self.temporal_transformer(m.transformer_in, h)
        self.pool.put(*h.parts)
        h = nh
        # ---- down (every layer output and every downsampled tensor is a skip connection)
        skips = [h]
        for blk in m.down_blocks:
            for i in range(len(blk.resnets)):
                h = self.layer(blk, i, h)
                skips.append(h)
            if blk.downsamplers is not None:
                h = self.conv(h, blk.downsamplers[0].conv, nt.GEMM_CONV3X3_S2)
                skips.append(h)
        # ---- mid (unet_3d_blocks.py:386-420)
        mid = m.mid_block
        h = self.resnet2d(mid.resnets[0], h)  # its input stays alive as the last skip
        h = self.temporal_conv_block(mid.temp_convs[0], h)
        for attn, tattn, resnet, tconv in zip(mid.attentions, mid.temp_attentions, mid.resnets[1:], mid.temp_convs[1:]):
            h = self._consume(self.spatial_transformer(attn, h), h)
            h = self._consume(self.temporal_transformer(tattn, h), h)
            h = self._consume(self.resnet2d(resnet, h), h)
            h = self.temporal_conv_block(tconv, h)
        # ---- up: virtual concat with the popped skip
        for blk in m.up_blocks:
            for i in range(len(blk.resnets)):
                skip = skips.pop()
                cat = Act([h.t, skip.t], h.n_img, h.h, h.w)
                nh = self.layer(blk, i, cat)
                self.pool.put(h.t, skip.t)
                h = nh
            if blk.upsamplers is not None:
                h = self._consume(self.conv(h, blk.upsamplers[0].conv, nt.GEMM_CONV3X3_UP2), h)
        # ---- out
        t = self.gn(h, m.conv_norm_out, B * F, H * W, True)
        y = self.conv(Act(t, h.n_img, h.h, h.w), m.conv_out, nt.GEMM_CONV3X3, out_dtype=torch.float32)
        ops.tokens_to_ncfhw(y.t, out)

    def _consume(self, new, old):
        self.pool.put(*old.parts)
        return new

    def layer(self, blk, i, h):
        """resnet -> temporal conv -> [spatial transformer -> temporal transformer] (unet_3d_blocks.py:547-561);
        the input belongs to the caller (skip connection / virtual concat)."""
        y = self.resnet2d(blk.resnets[i], h)
        y = self.temporal_conv_block(blk.temp_convs[i], y)
        if blk.has_cross_attention:
            y = self._consume(self.spatial_transformer(blk.attentions[i], y), y)
            y = self._consume(self.temporal_transformer(blk.temp_attentions[i], y), y)
        return y

    def resnet2d(self, rb, x):
        """diffusers ResnetBlock2D; x (possibly a virtual concat) is left to the caller."""
        B, F = self.B, self.F
        hw = x.h * x.w
        assert rb.output_scale_factor == 1.0
        off, cout = self.emb_off[id(rb)], rb.out_channels
        t = self.gn(x, rb.norm1, B * F, hw, True)
        h1 = self.conv(Act(t, x.n_img, x.h, x.w), rb.conv1, nt.GEMM_CONV3X3,
                       rowvec=self.emb_all[:, off:off + cout], rowvec_div=F * hw)
        self.pool.put(t)
        t2 = self.gn(h1, rb.norm2, B * F, hw, True)
        self.pool.put(h1.t)
        if rb.conv_shortcut is None:
            assert x.p1 is None
            skip, own = x.t, False
        else:
            skip = self.buf(x.M, cout)
            self.ops.gemm(x.parts[0], self.pk.mat(rb.conv_shortcut), skip, M=x.M, N=cout, a1=x.p1,
                          bias=self.pk.bias(rb.conv_shortcut))
            own = True
        h2 = self.conv(Act(t2, x.n_img, x.h, x.w), rb.conv2, nt.GEMM_CONV3X3, residual=skip)
        self.pool.put(t2)
        if own:
            self.pool.put(skip)
        return h2
