"""Native KL-VAE decode: all frames of a clip in one batched pass of the same gfx950 kernels the
UNet uses (implicit-GEMM 3x3 convs with folded nearest-x2 upsampling, two-phase GroupNorm+swish,
the mid AttnBlock as two batched GEMMs + a row softmax).  Replaces the per-frame Python loop of
``LatentDiffusion.decode_first_stage_2DAE`` (reference lvdm/models/ddpm3d.py:666-679) ->
``AutoencoderKL.decode`` (autoencoder.py:110-113) -> ``Decoder.forward`` (ae_modules.py:602-641)."""
import os

import torch

from . import native as nt
from .native import on_tensor_device
from .engine import Act, _Engine, leaf_out_channels
from .vae import AttnBlock


class VAEDecodeEngine(_Engine):
    def __init__(self, vae, ops):
        super().__init__(ops)
        self.vae = vae
        # GroupNorm statistics from the producing convs' epilogues (engine._Engine.gn): the decoder's 128-channel norms at
        # 320x512 are pure bandwidth.  The gradient engine (a subclass) keeps the standalone statistics it saves for its backward.
        if type(self).__name__ in ("VAEDecodeEngine", "VAEEncodeEngine"):
            self.fuse_gn = os.environ.get("T2V_FUSE_GN", "1") == "1"

    # The decoder's stride-1 3x3 convs (ae_modules.py:183-203: 128 / 256 / 512 channels over 40x64 ... 320x512 images, 25 TFLOP per
    # clip) are the halo-slab kernel's best case — large images, whole 10 x 32 tiles — but their widths are not multiples of its
    # 80-channel wave tiles.  The kernel pads the last channel tile (rows >= N are out-of-range DMA lanes = zeros, never stored):
    # 128 -> 160, 256 -> 320, 512 -> 560 / 640 columns of MFMA work.  T2V_VAE_HALO=0: the tuned t2v_gemm tiles (rounds 1-4).
    vae_halo = os.environ.get("T2V_VAE_HALO", "1") == "1"
    # conv_out (128 -> 3 channels) by the direct small-Cout kernel (csrc/elementwise.hip, t2v_conv3x3_small_cout) instead of an MFMA tile
    # that is 97 % padding.  MEASURED SLOWER on MI355X (round 5, same box, interleaved: decode 27.98 / 28.23 ms with it, 25.47 / 25.48 ms
    # without, profiles/r05_vae_decode_ab.jsonl): ~3.3 ms against 0.75 ms — the matrix cores multiply the padding at 1 PFLOP/s, while the
    # VALU form is bound by its own load -> 460-instruction chunk -> load chain at three waves per SIMD.  Opt-in (T2V_SMALL_COUT=1), kept
    # as a tested negative result.
    small_cout = os.environ.get("T2V_SMALL_COUT", "0") == "1"
    small_cout_min_tokens = 1 << 18

    def _halo_width_ok(self, N):
        return N % 80 == 0 or (self.vae_halo and N % 16 == 0 and N >= 64)

    @on_tensor_device
    def decode_frames(self, z, scale):
        """z (b, zc, t, h, w) -> (b, out_ch, t, 8h, 8w) in z.dtype; z is multiplied by ``scale`` first."""
        assert z.dim() == 5
        self._check_weights(self.vae)
        key = (tuple(z.shape), z.dtype, float(scale), z.device)
        plan = self.plans.get(key)
        if plan is None:
            plan = self._own(self._record(z, scale))
            self._keep_plan(key, plan)
        else:
            plan["static"]["z"].copy_(z)
            self._run(plan)
        return plan["out"].clone()

    def _record(self, z, scale):
        ops = self.ops
        self._begin(z.device)
        dec = self.vae.decoder
        b, zc, t, h, w = z.shape
        up = 2 ** (dec.num_resolutions - 1)
        st = {"z": z.detach().clone().contiguous()}
        out = torch.empty(b, leaf_out_channels(dec.conv_out), t, h * up, w * up, dtype=z.dtype, device=z.device)
        plan = {"static": st, "out": out, "runs": 0}

        def body():
            self._forward(st["z"], scale, out)

        if getattr(ops, "is_native", False):
            ops.init()
            ops.recording = []
            try:
                body()
            finally:
                plan["rec"] = ops.recording
                ops.recording = None
        else:
            plan["fn"] = body
            body()
        plan["pool_bytes"] = self.pool.bytes
        return plan

    def _forward(self, z, scale, out):
        ops, pk, vae = self.ops, self.pk, self.vae
        dec = vae.decoder
        b, zc, t, h, w = z.shape
        n_img = b * t
        assert zc <= 8
        zt = self.buf(n_img * h * w, zc)
        ops.ncfhw_to_tokens(z, zt)
        # post_quant_conv (1x1, zc->zc) as the centre tap of a direct 3x3 with cout padded to 8; the
        # 1/scale_factor of decode_first_stage_2DAE is folded into its weights
        pq = vae.post_quant_conv

        def pq_weights():
            wq = pq.weight.detach().float().reshape(pq.weight.shape[0], zc) * scale
            w3 = torch.zeros(8, 9, zc, dtype=torch.float32)
            w3[: wq.shape[0], 4, :] = wq.cpu()
            b8 = torch.zeros(8, dtype=torch.float32)
            b8[: wq.shape[0]] = pq.bias.detach().float().cpu()
            return w3.reshape(8, -1).to(self.device).contiguous(), b8.to(self.device)

        w3, b8 = pk._memo(("pq", id(pq), float(scale)), pq_weights)
        z8 = self.buf(n_img * h * w, 8)
        ops.conv_small(zt, n_img, h, w, w3, b8, z8)
        cin_w = pk.small_conv(dec.conv_in, cin_pad=8)
        h0 = self.buf(n_img * h * w, leaf_out_channels(dec.conv_in))
        ops.conv_small(z8, n_img, h, w, cin_w, pk.bias(dec.conv_in), h0)
        self.pool.put(zt, z8)
        x = Act(h0, n_img, h, w)
        x = self.resnet_block(dec.mid.block_1, x)
        if isinstance(dec.mid.attn_1, AttnBlock):
            x = self.attn_block(dec.mid.attn_1, x)
        x = self.resnet_block(dec.mid.block_2, x)
        for lvl in reversed(range(dec.num_resolutions)):
            for ib in range(dec.num_res_blocks + 1):
                x = self.resnet_block(dec.up[lvl].block[ib], x)
                if len(dec.up[lvl].attn) > 0:
                    x = self.attn_block(dec.up[lvl].attn[ib], x)
            if lvl != 0:
                ups = dec.up[lvl].upsample
                assert ups.with_conv
                nx = self.conv(x, ups.conv, nt.GEMM_CONV3X3_UP2)
                self.pool.put(x.t)
                x = nx
        tt = self.gn(x, dec.norm_out, n_img, x.h * x.w, True)
        self.pool.put(x.t)
        n_out = leaf_out_channels(dec.conv_out)
        if (self.small_cout and hasattr(ops, "conv_small_cout") and tt.dtype == torch.bfloat16 and n_img * x.h * x.w >= self.small_cout_min_tokens
                and ops.conv_small_cout_supported(x.w, tt.shape[1], n_out)):
            # conv_out (ae_modules.py:641: 128 -> 3 channels): a direct VALU conv instead of an MFMA tile that is 97 % padding
            yt = self.buf(n_img * x.h * x.w, n_out, torch.float32)
            ops.conv_small_cout(tt, n_img, x.h, x.w, pk.small_conv(dec.conv_out), pk.bias(dec.conv_out), yt)
        else:
            yt = self.conv(Act(tt, n_img, x.h, x.w), dec.conv_out, nt.GEMM_CONV3X3, out_dtype=torch.float32).t
        self.pool.put(tt)
        ops.tokens_to_ncfhw(yt, out)
        self.pool.put(yt)

    def resnet_block(self, rb, x):
        """ResnetBlock with temb=None (ae_modules.py:183-203); consumes (frees) its input."""
        hw = x.h * x.w
        t1 = self.gn(x, rb.norm1, x.n_img, hw, True, then=self.pk.conv(rb.conv1))
        h1 = self.conv(Act(t1, x.n_img, x.h, x.w), rb.conv1, nt.GEMM_CONV3X3)
        self.pool.put(t1)
        t2 = self.gn(h1, rb.norm2, x.n_img, hw, True, then=self.pk.conv(rb.conv2))
        self.pool.put(h1.t)
        skip, own = x.t, False
        if rb.in_channels != rb.out_channels:
            if rb.use_conv_shortcut:
                skip = self.conv(x, rb.conv_shortcut, nt.GEMM_CONV3X3).t
            else:
                skip = self.linear(x.t, rb.nin_shortcut)
            own = True
        h2 = self.conv(Act(t2, x.n_img, x.h, x.w), rb.conv2, nt.GEMM_CONV3X3, residual=skip)
        self.pool.put(t2, x.t)
        if own:
            self.pool.put(skip)
        return h2

    def attn_block(self, ab, x):
        """AttnBlock (ae_modules.py:48-73): S = q k^T / sqrt(c) -> softmax -> S v, per image."""
        ops, pk = self.ops, self.pk
        c, seq, n_img = x.C, x.h * x.w, x.n_img
        kp = ((seq + 63) // 64) * 64
        t = self.gn(x, ab.norm, n_img, seq, False)
        q = self.linear(t, ab.q)
        k = self.linear(t, ab.k)
        vt = self.buf(n_img * c, kp)
        if kp != seq:
            ops.fill_zero(vt)
        # V^T per image (bias of v is added after the PV product: softmax rows sum to 1)
        ops.gemm(pk.mat(ab.v), t, vt, M=c, N=seq, batch=n_img, w_strides=(seq * t.stride(0), 0), o_strides=(c * kp, 0))
        s = self.buf(n_img * seq, kp)
        ops.gemm(q, k, s, M=seq, N=seq, alpha=float(int(c) ** -0.5), batch=n_img, a_strides=(seq * q.stride(0), 0),
                 w_strides=(seq * k.stride(0), 0), o_strides=(seq * kp, 0))
        ops.softmax_rows(s, n_img * seq, seq, kp, kp)
        o = self.buf(n_img * seq, c)
        ops.gemm(s, vt, o, M=seq, N=c, batch=n_img, a_strides=(seq * kp, 0), w_strides=(c * kp, 0),
                 o_strides=(seq * c, 0), bias=pk.bias(ab.v))
        out = self.linear(o, ab.proj_out, residual=x.t, want_cs=True)
        cs = self.last_cs
        self.pool.put(t, q, k, vt, s, o, x.t)
        return Act(out, n_img, x.h, x.w, cs=[cs])


class VAEEncodeEngine(VAEDecodeEngine):
    """Native KL-VAE encode up to the posterior parameters: the step right before the
    hot path in v1 training (train_t2v_turbo_v1_lora.py:957-971).  Same kernels as the decoder plus the
    stride-2 conv with right/bottom padding (T2V_GEMM_CONV3X3_S2_PAD01); ``quant_conv`` (1x1) is folded into
    ``conv_out`` exactly (a pointwise map after the conv)."""

    @on_tensor_device
    def encode_frames(self, x):
        """x (b, 3, t, H, W) -> moments (b, 2*embed, t, H/8, W/8), fp32."""
        assert x.dim() == 5
        self._check_weights(self.vae)
        key = ("enc", tuple(x.shape), x.dtype, x.device)
        plan = self.plans.get(key)
        if plan is None:
            plan = self._own(self._record_enc(x))
            self._keep_plan(key, plan)
        else:
            plan["static"]["x"].copy_(x)
            self._run(plan)
        return plan["out"].clone()

    def _record_enc(self, x):
        ops = self.ops
        self._begin(x.device)
        enc = self.vae.encoder
        b, c, t, H, W = x.shape
        down = 2 ** (enc.num_resolutions - 1)
        st = {"x": x.detach().clone().contiguous()}
        out = torch.empty(b, self.vae.quant_conv.weight.shape[0], t, H // down, W // down, dtype=torch.float32, device=x.device)
        plan = {"static": st, "out": out, "runs": 0}

        def body():
            self._forward_enc(st["x"], out)

        if getattr(ops, "is_native", False):
            ops.init()
            ops.recording = []
            try:
                body()
            finally:
                plan["rec"] = ops.recording
                ops.recording = None
        else:
            plan["fn"] = body
            body()
        plan["pool_bytes"] = self.pool.bytes
        return plan

    def _forward_enc(self, x, out):
        ops, pk, vae = self.ops, self.pk, self.vae
        enc = vae.encoder
        b, c, t, H, W = x.shape
        n_img = b * t
        assert c <= 4
        xt = self.buf(n_img * H * W, 4)
        if c < 4:
            ops.fill_zero(xt)  # the padded input channel meets zero weights, but must be finite
        ops.ncfhw_to_tokens(x, xt)
        h0 = self.buf(n_img * H * W, leaf_out_channels(enc.conv_in))
        ops.conv_small(xt, n_img, H, W, pk.small_conv(enc.conv_in, cin_pad=4), pk.bias(enc.conv_in), h0)
        self.pool.put(xt)
        a = Act(h0, n_img, H, W)
        for lvl in range(enc.num_resolutions):
            for ib in range(enc.num_res_blocks):
                a = self.resnet_block(enc.down[lvl].block[ib], a)
                if len(enc.down[lvl].attn) > 0:
                    a = self.attn_block(enc.down[lvl].attn[ib], a)
            if lvl != enc.num_resolutions - 1:
                ds = enc.down[lvl].downsample
                assert ds.with_conv
                w = pk.conv(ds.conv)
                ho, wo = (a.h - 2) // 2 + 1, (a.w - 2) // 2 + 1
                o = self.buf(n_img * ho * wo, w.shape[0])
                ops.gemm(a.t, w, o, M=n_img * ho * wo, N=w.shape[0], mode=nt.GEMM_CONV3X3_S2_PAD01, n_img=n_img, h=a.h,
                         wd=a.w, bias=pk.bias(ds.conv))
                self.pool.put(a.t)
                a = Act(o, n_img, ho, wo)
        a = self.resnet_block(enc.mid.block_1, a)
        if isinstance(enc.mid.attn_1, AttnBlock):
            a = self.attn_block(enc.mid.attn_1, a)
        a = self.resnet_block(enc.mid.block_2, a)
        tt = self.gn(a, enc.norm_out, n_img, a.h * a.w, True)
        self.pool.put(a.t)

        def folded():  # quant_conv o conv_out
            wq = vae.quant_conv.weight.detach().float().flatten(1)                       # [2e, 2z]
            wc = enc.conv_out.weight.detach().float().permute(0, 2, 3, 1).flatten(1)     # [2z, 9*C] tap-major
            wf = (wq @ wc).to(self.device, self.adt).contiguous()
            bf = (wq @ enc.conv_out.bias.detach().float() + vae.quant_conv.bias.detach().float()).to(self.device).contiguous()
            return wf, bf

        wf, bf = pk._memo(("enc_out", id(enc.conv_out), id(vae.quant_conv)), folded)
        mom = self.buf(n_img * a.h * a.w, wf.shape[0], torch.float32)
        ops.gemm(tt, wf, mom, M=n_img * a.h * a.w, N=wf.shape[0], mode=nt.GEMM_CONV3X3, n_img=n_img, h=a.h, wd=a.w, bias=bf)
        self.pool.put(tt)
        ops.tokens_to_ncfhw(mom, out)
        self.pool.put(mom)
