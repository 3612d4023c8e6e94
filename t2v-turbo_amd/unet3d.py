"""VideoCrafter2 3D-UNet behind the reference's interface.

Mirrors (names, ctor kwargs, child registration order, state-dict keys) the reference classes of
``lvdm/modules/networks/openaimodel3d.py`` and ``lvdm/modules/attention.py`` so that checkpoints,
``utils/lora.py`` injection (exact-class ``nn.Linear/Conv2d/Conv3d`` leaves under a class named
``UNetModel``), ``named_modules()`` probes (``output_blocks.N.2.transformer_blocks.0.attn1``) and
the training / sampling scripts work unchanged — including the misspelt ``temopral_conv`` key.

Two execution paths, selected by where the input lives:
  * CUDA tensor, no autograd  -> the gfx950 HIP engine (``engine.UNetEngine``): token-major bf16
    activations, hand-written kernels through the C-ABI, hipGraph replay.  No fallback: a missing
    library or unsupported shape raises.
  * CPU tensor (or autograd)  -> each module's ``forward`` below: plain torch ops with the
    reference's semantics ("plumbing" configuration C1 and the autograd path of training).
"""
import os
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from .nn_util import EngineBox, checkpoint_call, sinusoidal_embedding


class GroupNormSpecific(nn.GroupNorm):
    """GroupNorm that returns the input dtype (lvdm/basics.py:78-81)."""

    def forward(self, x):
        return super().forward(x).type(x.dtype)


def normalization(channels, num_groups=32):
    return GroupNormSpecific(num_groups, channels)


def zero_module(m):
    for p in m.parameters():
        p.detach().zero_()
    return m


# =================================================================================== attention
class CrossAttention(nn.Module):
    """q/k/v projections + softmax(QK^T/sqrt(d))V + out projection (attention.py:50-164)."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0, relative_position=False,
                 temporal_length=None, img_cross_attention=False, record_attn_probs=False):
        super().__init__()
        if relative_position or img_cross_attention:
            raise NotImplementedError("relative_position / img_cross_attention are unused by t2v-turbo configs")
        inner = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))
        self.relative_position = False
        self.img_cross_attention = False
        self.record_attn_probs = record_attn_probs
        self.attention_probs = None

    def forward(self, x, context=None, mask=None):
        ctx = x if context is None else context
        q, k, v = self.to_q(x), self.to_k(ctx), self.to_v(ctx)
        b, n, _ = q.shape
        h, d = self.heads, self.dim_head

        def heads_first(t):
            return t.reshape(t.shape[0], t.shape[1], h, d).permute(0, 2, 1, 3).reshape(t.shape[0] * h, t.shape[1], d)

        q, k, v = heads_first(q), heads_first(k), heads_first(v)
        sim = torch.bmm(q, k.transpose(1, 2)) * self.scale
        if mask is not None:
            mask = mask.repeat_interleave(h, dim=0)
            sim = sim.masked_fill(~(mask > 0.5), -torch.finfo(sim.dtype).max)
        probs = sim.softmax(dim=-1)
        if self.record_attn_probs:
            self.attention_probs = probs
        out = torch.bmm(probs, v).reshape(b, h, n, d).permute(0, 2, 1, 3).reshape(b, n, h * d)
        return self.to_out(out)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        a, gate = self.proj(x).chunk(2, dim=-1)
        return a * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.0):
        super().__init__()
        inner = int(dim * mult)
        dim_out = dim if dim_out is None else dim_out
        first = GEGLU(dim, inner) if glu else nn.Sequential(nn.Linear(dim, inner), nn.GELU())
        self.net = nn.Sequential(first, nn.Dropout(dropout), nn.Linear(inner, dim_out))

    def forward(self, x):
        return self.net(x)


class BasicTransformerBlock(nn.Module):
    """pre-LN: x += attn1(LN x); x += attn2(LN x, ctx); x += ff(LN x) (attention.py:243-311)."""

    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False, attention_cls=None, img_cross_attention=False, record_attn_probs=False):
        super().__init__()
        cls = CrossAttention if attention_cls is None else attention_cls
        self.disable_self_attn = disable_self_attn
        self.attn1 = cls(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout,
                         context_dim=context_dim if disable_self_attn else None,
                         record_attn_probs=record_attn_probs)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = cls(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout,
                         img_cross_attention=img_cross_attention)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)
        self.checkpoint = checkpoint

    def forward(self, x, context=None, mask=None):
        fn = partial(self._forward, mask=mask) if mask is not None else self._forward
        args = (x,) if (context is None or mask is not None) else (x, context)
        return checkpoint_call(fn, args, self.checkpoint)

    def _forward(self, x, context=None, mask=None):
        x = self.attn1(self.norm1(x), context=context if self.disable_self_attn else None, mask=mask) + x
        x = self.attn2(self.norm2(x), context=context, mask=mask) + x
        return self.ff(self.norm3(x)) + x


class SpatialTransformer(nn.Module):
    """GroupNorm -> tokens (h w) -> proj_in -> blocks -> proj_out -> + input (attention.py:314-389)."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None, use_checkpoint=True,
                 disable_self_attn=False, use_linear=False, img_cross_attention=False):
        super().__init__()
        self.in_channels = in_channels
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner) if use_linear else nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim,
                                  img_cross_attention=img_cross_attention, disable_self_attn=disable_self_attn,
                                  checkpoint=use_checkpoint) for _ in range(depth)])
        self.proj_out = zero_module(nn.Linear(inner, in_channels) if use_linear else nn.Conv2d(inner, in_channels, 1))
        self.use_linear = use_linear

    def forward(self, x, context=None):
        b, c, h, w = x.shape
        y = self.norm(x)
        if not self.use_linear:
            y = self.proj_in(y)
        y = y.flatten(2).transpose(1, 2)
        if self.use_linear:
            y = self.proj_in(y)
        for blk in self.transformer_blocks:
            y = blk(y, context=context)
        if self.use_linear:
            y = self.proj_out(y)
        y = y.transpose(1, 2).reshape(b, -1, h, w)
        if not self.use_linear:
            y = self.proj_out(y)
        return y + x


class TemporalTransformer(nn.Module):
    """Same block structure, sequence = frames of one pixel (attention.py:392-513).  Only the
    t2v-turbo configuration (only_self_att, no causal mask, no relative position) is supported."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None, use_checkpoint=True,
                 use_linear=False, only_self_att=True, causal_attention=False, relative_position=False,
                 temporal_length=None, record_attn_probs=False):
        super().__init__()
        if not only_self_att or causal_attention or relative_position:
            raise NotImplementedError("TemporalTransformer: only temporal_selfatt_only=True, non-causal, "
                                      "no relative position (the VideoCrafter2 / t2v-turbo config) is supported")
        self.only_self_att = True
        self.relative_position = False
        self.causal_attention = False
        self.in_channels = in_channels
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner) if use_linear else nn.Conv1d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=None, checkpoint=use_checkpoint,
                                  record_attn_probs=record_attn_probs) for _ in range(depth)])
        self.proj_out = zero_module(nn.Linear(inner, in_channels) if use_linear else nn.Conv1d(inner, in_channels, 1))
        self.use_linear = use_linear

    @staticmethod
    def _pointwise(proj, y):
        """Linear (possibly LoRA-injected) or the k=1 Conv1d of init_attn: a per-token matmul either way."""
        if isinstance(proj, nn.Conv1d):
            return F.linear(y, proj.weight.reshape(proj.weight.shape[0], -1), proj.bias)
        return proj(y)

    def forward(self, x, context=None):
        b, c, t, h, w = x.shape
        y = self.norm(x).permute(0, 3, 4, 2, 1).reshape(b * h * w, t, c)  # (b h w) t c
        y = self._pointwise(self.proj_in, y)
        for blk in self.transformer_blocks:
            y = blk(y)
        y = self._pointwise(self.proj_out, y)
        return y.reshape(b, h, w, t, c).permute(0, 4, 3, 1, 2) + x


# =================================================================================== conv blocks
class TimestepBlock(nn.Module):
    """Marker: forward(x, emb, batch_size)."""


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """Routes (emb | context | frame-unfolded tensor) to each child by type (openaimodel3d.py:25-45)."""

    def forward(self, x, emb, context=None, batch_size=None):
        for layer in self:
            if isinstance(layer, TimestepBlock):
                x = layer(x, emb, batch_size)
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context)
            elif isinstance(layer, TemporalTransformer):
                n, c, h, w = x.shape
                x5 = x.reshape(batch_size, n // batch_size, c, h, w).transpose(1, 2)
                x = layer(x5, context).transpose(1, 2).reshape(n, c, h, w)
            else:
                x = layer(x)
        return x


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert dims == 2, "only dims=2 is used by the VideoCrafter2 UNet"
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.dims = dims
        if use_conv:
            self.op = nn.Conv2d(self.channels, self.out_channels, 3, stride=2, padding=padding)
        else:
            assert self.channels == self.out_channels
            self.op = nn.AvgPool2d(kernel_size=2, stride=2)

    def forward(self, x):
        assert x.shape[1] == self.channels
        return self.op(x)


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert dims == 2, "only dims=2 is used by the VideoCrafter2 UNet"
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.dims = dims
        if use_conv:
            self.conv = nn.Conv2d(self.channels, self.out_channels, 3, padding=padding)

    def forward(self, x):
        assert x.shape[1] == self.channels
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        return self.conv(x) if self.use_conv else x


class TemporalConvBlock(nn.Module):
    """4 x [GroupNorm(all frames) -> SiLU -> (Dropout) -> Conv3d(3,1,1)] + identity; the last conv
    starts at zero (openaimodel3d.py:257-309)."""

    def __init__(self, in_channels, out_channels=None, dropout=0.0, spatial_aware=False):
        super().__init__()
        if spatial_aware:
            raise NotImplementedError("tempspatial_aware convs are not used by t2v-turbo")
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels

        def stage(cin, cout, with_dropout):
            layers = [nn.GroupNorm(32, cin), nn.SiLU()]
            if with_dropout:
                layers.append(nn.Dropout(dropout))
            layers.append(nn.Conv3d(cin, cout, (3, 1, 1), padding=(1, 0, 0)))
            return nn.Sequential(*layers)

        self.conv1 = stage(in_channels, out_channels, False)
        self.conv2 = stage(out_channels, in_channels, True)
        self.conv3 = stage(out_channels, in_channels, True)
        self.conv4 = stage(out_channels, in_channels, True)
        nn.init.zeros_(self.conv4[-1].weight)
        nn.init.zeros_(self.conv4[-1].bias)

    def forward(self, x):
        return self.conv4(self.conv3(self.conv2(self.conv1(x)))) + x


class ResBlock(TimestepBlock):
    """GN-SiLU-conv, + time embedding, GN-SiLU-conv(zero), + skip, then the temporal conv block
    (openaimodel3d.py:115-254).  resblock_updown / scale-shift-norm variants are not built by the
    t2v-turbo config and are rejected."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_scale_shift_norm=False, dims=2,
                 use_checkpoint=False, use_conv=False, up=False, down=False, use_temporal_conv=False,
                 tempspatial_aware=False):
        super().__init__()
        if use_scale_shift_norm or up or down or dims != 2:
            raise NotImplementedError("ResBlock: scale-shift norm / up / down / dims!=2 are not part of the "
                                      "VideoCrafter2 configuration")
        self.channels = channels
        self.emb_channels = emb_channels
        self.dropout = dropout
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.use_checkpoint = use_checkpoint
        self.use_scale_shift_norm = False
        self.use_temporal_conv = use_temporal_conv
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(),
                                       nn.Conv2d(channels, self.out_channels, 3, padding=1))
        self.updown = False
        self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 3, padding=1)
        else:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 1)
        if use_temporal_conv:
            self.temopral_conv = TemporalConvBlock(self.out_channels, self.out_channels, dropout=0.1,
                                                   spatial_aware=tempspatial_aware)

    def forward(self, x, emb, batch_size=None):
        fn = partial(self._forward, batch_size=batch_size) if batch_size else self._forward
        return checkpoint_call(fn, (x, emb), self.use_checkpoint)

    def _forward(self, x, emb, batch_size=None):
        h = self.in_layers(x)
        e = self.emb_layers(emb).type(h.dtype)
        h = self.out_layers(h + e[:, :, None, None])
        h = self.skip_connection(x) + h
        if self.use_temporal_conv and batch_size:
            n, c, hh, ww = h.shape
            h5 = h.reshape(batch_size, n // batch_size, c, hh, ww).transpose(1, 2)
            h = self.temopral_conv(h5).transpose(1, 2).reshape(n, c, hh, ww)
        return h


# =================================================================================== the UNet
_WARNED_ATEN = set()


def _warn_aten_route(why):
    """The torch composite path on a CUDA tensor is a fallback (ATen kernels, ~3x slower than the native engines): say so once
    per reason."""
    if why not in _WARNED_ATEN:
        _WARNED_ATEN.add(why)
        import warnings
        warnings.warn(f"t2v_turbo_amd UNetModel: running the torch (ATen) path on a CUDA tensor — {why}; "
                      'set native_mode = "off" to silence, see INTEGRATION.md', RuntimeWarning, stacklevel=3)


def _flat_grad_buffer(eng, leaves):
    """The ``dist.FlatGradSync`` whose flat fp32 buffer's slots ARE the LoRA tensors' ``.grad`` (same order and
    offsets as ``bind_lora``), or None.  Only a buffer that announced itself (``FlatGradSync`` tags its parameters) is taken:
    writing into ``.grad`` bypasses AccumulateGrad, so a look-alike layout owned by somebody else — DDP's
    ``gradient_as_bucket_view`` bucket, accelerate's reducer — or any parameter with gradient hooks gets its gradients
    through autograd instead (the hooks then fire as usual)."""
    sync = getattr(eng.lora_params[0], "_t2v_flat_sync", None)
    if sync is None or sync.flat.dtype != torch.float32 or sync.numel != eng.lora_numel:
        return None
    base = sync.flat.data_ptr()
    for p in eng.lora_params:
        g = p.grad
        if getattr(p, "_t2v_flat_sync", None) is not sync or g is None or g.dtype != torch.float32 or not g.is_contiguous() \
                or g.data_ptr() != base + 4 * eng.lora_off[id(p)]:
            return None
        if p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None):
            return None
    return sync


class _NativeStudent(torch.autograd.Function):
    """UNet forward whose backward — d/d(latents), d/d(emb_all) and every token-row LoRA weight gradient — runs on the native
    gradient engine.  Inputs: latents, the conditioning branch's output (torch keeps differentiating behind it), and the
    LoRA tensors the engine owns (so autograd routes their gradients to the optimizer's ``.grad`` slots)."""

    @staticmethod
    def forward(ctx, x, emb_all, model, args, *leaves):
        eng = model.native_train_engine()
        timesteps, context, fps, tc, mc = args
        y = eng.forward_tape(x.detach(), timesteps, context.detach(), fps, tc, mc, emb_all=emb_all.detach())
        ctx.model, ctx.plan, ctx.fwd_id = model, eng._last, eng._last["fwd_id"]
        ctx.leaves = leaves
        ctx.x_dtype, ctx.e_dtype = x.dtype, emb_all.dtype
        return y

    @staticmethod
    def backward(ctx, dout):
        eng = ctx.model.native_train_engine()
        if ctx.plan["fwd_id"] != ctx.fwd_id:
            raise RuntimeError("native student: another grad-mode forward of the same shape ran before this backward "
                               "(the engine keeps one outstanding tape per input shape)")
        eng._last = ctx.plan
        sync = _flat_grad_buffer(eng, ctx.leaves)
        if sync is not None:
            # every LoRA tensor's .grad is already its slot of ONE flat fp32 buffer in bind_lora order (dist.FlatGradSync): the engine
            # adds its weight gradients there in one launch — what 1096 AccumulateGrad nodes would do with 1096 small kernels
            dx = eng.backward(dout, flat_grad=sync.flat, accumulate=True, grad_sync=sync)
            return (dx.to(ctx.x_dtype), eng.d_emb_all.to(ctx.e_dtype).clone(), None, None, *([None] * len(ctx.leaves)))
        flat = torch.empty(eng.lora_numel, dtype=torch.float32, device=dout.device)  # fresh: .grad may keep views of it
        dx = eng.backward(dout, flat_grad=flat, accumulate=False)
        grads = []
        for p in ctx.leaves:
            o = eng.lora_off[id(p)]
            grads.append(flat[o:o + p.numel()].view_as(p).to(p.dtype))
        return (dx.to(ctx.x_dtype), eng.d_emb_all.to(ctx.e_dtype).clone(), None, None, *grads)


class _NativeStudentFull(torch.autograd.Function):
    """UNet forward of the FULL fine-tuning student (train_latent_t2v_turbo_v2.py:669,798-816,1262: every parameter trainable, no LoRA):
    d/d(latents), d/d(emb_all) and the gradient of every parameter outside the B-row conditioning branch come from the native gradient
    engine (engine_full.py); torch keeps differentiating the conditioning branch behind ``emb_all``."""

    @staticmethod
    def forward(ctx, x, emb_all, model, args, *params):
        eng = model.native_full_engine()
        timesteps, context, fps, tc, mc = args
        y = eng.forward_tape(x.detach(), timesteps, context.detach(), fps, tc, mc, emb_all=emb_all.detach())
        ctx.model, ctx.plan, ctx.fwd_id = model, eng._last, eng._last["fwd_id"]
        ctx.params = params
        ctx.x_dtype, ctx.e_dtype = x.dtype, emb_all.dtype
        return y

    @staticmethod
    def backward(ctx, dout):
        eng = ctx.model.native_full_engine()
        if ctx.plan["fwd_id"] != ctx.fwd_id:
            raise RuntimeError("native student: another grad-mode forward of the same shape ran before this backward "
                               "(the engine keeps one outstanding tape per input shape)")
        eng._last = ctx.plan
        dx = eng.backward(dout)
        grads = [None if (g is None or not p.requires_grad) else g.to(p.dtype) for p, g in zip(ctx.params, eng.full_grads(ctx.params))]
        return (dx.to(ctx.x_dtype), eng.d_emb_all.to(ctx.e_dtype).clone(), None, None, *grads)


class UNetModel(nn.Module):
    """Same constructor signature as the reference (openaimodel3d.py:340-374)."""

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0.0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, context_dim=None, use_scale_shift_norm=False,
                 resblock_updown=False, num_heads=-1, num_head_channels=-1, transformer_depth=1, use_linear=False,
                 use_checkpoint=False, temporal_conv=False, tempspatial_aware=False, temporal_attention=True,
                 temporal_selfatt_only=True, use_relative_position=True, use_causal_attention=False,
                 temporal_length=None, use_fp16=False, addition_attention=False, use_image_attention=False,
                 temporal_transformer_depth=1, fps_cond=False, time_cond_proj_dim=None, motion_cond_proj_dim=None,
                 record_attn_probs=False):
        super().__init__()
        if num_heads == -1:
            assert num_head_channels != -1, "Either num_heads or num_head_channels has to be set"
        if num_head_channels == -1:
            assert num_heads != -1, "Either num_heads or num_head_channels has to be set"
        if resblock_updown or use_scale_shift_norm or dims != 2 or use_image_attention:
            raise NotImplementedError("UNetModel: option outside the VideoCrafter2 / t2v-turbo configuration")
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = attention_resolutions
        self.dropout = dropout
        self.channel_mult = channel_mult
        self.conv_resample = conv_resample
        self.temporal_attention = temporal_attention
        self.use_checkpoint = use_checkpoint
        self.dtype = torch.float16 if use_fp16 else torch.float32
        self.addition_attention = addition_attention
        self.use_image_attention = use_image_attention
        self.fps_cond = fps_cond
        self.time_cond_proj_dim = time_cond_proj_dim
        self.motion_cond_proj_dim = motion_cond_proj_dim
        mc, ted = model_channels, model_channels * 4

        def mlp():
            return nn.Sequential(nn.Linear(mc, ted), nn.SiLU(), nn.Linear(ted, ted))

        self.time_embed = mlp()
        if fps_cond:
            self.fps_embedding = mlp()
        self.time_cond_proj = nn.Linear(time_cond_proj_dim, mc, bias=False) if time_cond_proj_dim is not None else None
        if motion_cond_proj_dim is not None:
            self.motion_cond_proj = nn.Linear(motion_cond_proj_dim, mc, bias=False)
            self.combine_proj = nn.Linear(mc * 2, mc, bias=False)
        else:
            self.motion_cond_proj = None
            self.combine_proj = None

        def res(cin, cout):
            return ResBlock(cin, ted, dropout, out_channels=cout, dims=dims, use_checkpoint=use_checkpoint,
                            use_scale_shift_norm=False, tempspatial_aware=tempspatial_aware,
                            use_temporal_conv=temporal_conv)

        def head_split(ch):
            return (num_heads, ch // num_heads) if num_head_channels == -1 else (ch // num_head_channels, num_head_channels)

        def temporal(ch, heads, dh, depth, record=False):
            return TemporalTransformer(ch, heads, dh, depth=depth, context_dim=context_dim, use_linear=use_linear,
                                       use_checkpoint=use_checkpoint, only_self_att=temporal_selfatt_only,
                                       causal_attention=use_causal_attention, relative_position=use_relative_position,
                                       temporal_length=temporal_length, record_attn_probs=record)

        def attention_pair(ch, record=False):
            heads, dh = head_split(ch)
            out = [SpatialTransformer(ch, heads, dh, depth=transformer_depth, context_dim=context_dim,
                                      use_linear=use_linear, use_checkpoint=use_checkpoint, disable_self_attn=False,
                                      img_cross_attention=False)]
            if temporal_attention:
                out.append(temporal(ch, heads, dh, temporal_transformer_depth, record))
            return out

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(in_channels, mc, 3, padding=1))])
        if addition_attention:
            # NB: 8 heads regardless of num_head_channels, Conv1d projections (use_linear is not forwarded)
            self.init_attn = TimestepEmbedSequential(TemporalTransformer(
                mc, n_heads=8, d_head=num_head_channels, depth=transformer_depth, context_dim=context_dim,
                use_checkpoint=use_checkpoint, only_self_att=temporal_selfatt_only,
                causal_attention=use_causal_attention, relative_position=use_relative_position,
                temporal_length=temporal_length))
        skip_chans = [mc]
        ch, ds = mc, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, mult * mc)]
                ch = mult * mc
                if ds in attention_resolutions:
                    layers += attention_pair(ch)
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                skip_chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims, out_channels=ch)))
                skip_chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(res(ch, ch), *attention_pair(ch), res(ch, ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [res(ch + skip_chans.pop(), mult * mc)]
                ch = mc * mult
                if ds in attention_resolutions:
                    layers += attention_pair(ch, record=record_attn_probs)
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), nn.SiLU(), zero_module(nn.Conv2d(mc, out_channels, 3, padding=1)))
        self._engine_box = EngineBox()

    # ------------------------------------------------------------------------------------------
    def forward(self, x, timesteps, context=None, features_adapter=None, fps=16, timestep_cond=None,
                motion_cond=None, **kwargs):
        if motion_cond is not None:
            assert timestep_cond is not None
        mode = getattr(self, "native_mode", "auto")
        if mode == "train":  # force the native gradient engine (raises where it cannot run)
            from .nn_util import walk_modules
            from .engine import is_lora_leaf
            if not any(is_lora_leaf(mod) for mod in walk_modules(self)) and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                return self._forward_native_full(x, timesteps, context, fps, timestep_cond, motion_cond)
            return self._forward_native_train(x, timesteps, context, fps, timestep_cond, motion_cond)
        if mode != "off" and x.is_cuda:  # "off": always the torch path
            route, why = self._auto_route(x, context, timestep_cond, features_adapter)
            if route == "infer":
                return self.native_engine()(x, timesteps, context, fps, timestep_cond, motion_cond)
            if route == "train":
                return self._forward_native_train(x, timesteps, context, fps, timestep_cond, motion_cond)
            if route == "train_full":
                return self._forward_native_full(x, timesteps, context, fps, timestep_cond, motion_cond)
            _warn_aten_route(why)
        return self._forward_composite(x, timesteps, context, features_adapter, fps, timestep_cond, motion_cond)

    def _auto_route(self, x, context, timestep_cond, features_adapter):
        """Which engine a CUDA call lands on (``native_mode = "auto"``):
          * no gradient wanted, no active dropout            -> "infer": inference engine (LoRA branches merged at pack time);
          * LoRA-injected student, only LoRA tensors trainable, with or without grad, train or eval mode
            (the student and target forwards of train_t2v_turbo_v1_lora.py:1022-1028,1161-1168)
                                                             -> "train": gradient engine (un-merged LoRA branch, counter-based dropout);
          * no gradient wanted, no LoRA, train mode with only the TemporalConvBlock dropouts live (the v1 teacher, which the
            reference never puts in eval mode)               -> "infer": inference engine + counter-based dropout masks;
          * no LoRA, gradients wanted and at least one parameter trainable — FULL fine-tuning, the student of
            train_latent_t2v_turbo_v2.py:669,798-816,1262 (train mode with the TemporalConvBlock dropouts live, or eval)
                                                             -> "train_full": gradient engine with base-weight gradients (engine_full.py);
          * anything else (gradients w.r.t. the context, input gradients of a frozen network through this route, adapters, other live
            dropouts)                                        -> the torch composite path, with a one-time warning."""
        from .nn_util import walk_modules, walk_parameters
        grad = torch.is_grad_enabled() and self._needs_grad(x, context, timestep_cond)
        mods = walk_modules(self) if (self.training or grad) else ()
        dropping = self.training and any(isinstance(mod, nn.Dropout) and mod.p > 0 for mod in mods)
        if features_adapter is not None:
            if not grad and not dropping:
                raise NotImplementedError("features_adapter is not used by t2v-turbo and not supported natively")
            return "composite", "features_adapter"
        if not grad and not dropping:
            return "infer", None
        from .engine import is_lora_leaf
        if not grad and not any(is_lora_leaf(mod) for mod in mods) and self._only_tconv_dropouts(mods):
            # a frozen, LoRA-free network left in train mode under no_grad: the v1 distillation teacher as the reference runs it
            # (train_t2v_turbo_v1_lora.py:621-626 never calls .eval(); forwards at :1105-1134) — inference dataflow + the
            # TemporalConvBlock dropouts as counter-based masks
            return "infer", None
        lora_ids = {id(w) for mod in mods if is_lora_leaf(mod) for w in (mod.lora_up.weight, mod.lora_down.weight)}
        if not lora_ids:
            if (grad and self.native_full and context is not None and any(p.requires_grad for p in walk_parameters(self))
                    and not (context.requires_grad or (timestep_cond is not None and timestep_cond.requires_grad))
                    and (not self.training or self._only_tconv_dropouts(mods))):
                return "train_full", None
            return "composite", ("a train-mode network with active Dropout and no LoRA" if not grad else
                                 "gradients without LoRA injection that the native full fine-tuning route does not take (input gradients only, "
                                 "gradients w.r.t. the context, other live dropouts, T2V_NATIVE_FULL=0)")
        if context is None:
            return "composite", "no text context"
        if any(p.requires_grad and id(p) not in lora_ids for p in walk_parameters(self)):
            return "composite", "trainable parameters besides the LoRA tensors"
        if grad and (context.requires_grad or (timestep_cond is not None and timestep_cond.requires_grad)):
            return "composite", "gradients w.r.t. the context / guidance embedding"
        return "train", None

    @staticmethod
    def _only_tconv_dropouts(mods):
        """Every active Dropout(p > 0) sits in a TemporalConvBlock stage (the only ones the inference engine applies)."""
        known = set()
        for blk in mods:
            if isinstance(blk, TemporalConvBlock):
                for stg in (blk.conv1, blk.conv2, blk.conv3, blk.conv4):
                    known.update(id(layer) for layer in stg if isinstance(layer, nn.Dropout))
        return all(id(mod) in known for mod in mods if isinstance(mod, nn.Dropout) and mod.p > 0 and mod.training)

    def _needs_grad(self, *tensors):
        if any(t is not None and t.requires_grad for t in tensors):
            return True
        return any(p.requires_grad for p in self.parameters())

    def native_engine(self):
        if self._engine_box.engine is None:
            from .engine import UNetEngine
            from .native import HipOps
            self._engine_box.engine = UNetEngine(self, HipOps())
        return self._engine_box.engine

    # ---- native FULL fine-tuning (every parameter trainable, no LoRA: train_latent_t2v_turbo_v2.py) ---------------------------------
    native_full = os.environ.get("T2V_NATIVE_FULL", "1") == "1"

    def native_full_engine(self):
        if getattr(self._engine_box, "full", None) is None:
            from .engine_unet_bwd import UNetGradEngine
            make_ops = getattr(self, "_native_ops_factory", None)  # tests substitute the emulated backend
            if make_ops is None:
                from .native import HipOps as make_ops
            eng = UNetGradEngine(self, make_ops())
            eng.bind_full(eng.engine_parameters(self))
            self._engine_box.full = eng
        return self._engine_box.full

    def _forward_native_full(self, x, timesteps, context, fps, timestep_cond, motion_cond):
        if not (x.is_cuda or getattr(self, "_native_ops_factory", None) is not None):
            raise RuntimeError("native full fine-tuning needs CUDA tensors")
        if context is None:
            raise ValueError("native full fine-tuning needs the text context")
        if context.requires_grad or (timestep_cond is not None and timestep_cond.requires_grad):
            raise RuntimeError("native full fine-tuning: gradients flow to the latents and the parameters only")
        eng = self.native_full_engine()
        emb_all = self.conditioning_emb_all(timesteps, fps, timestep_cond, motion_cond)
        args = (timesteps, context, fps, timestep_cond, motion_cond)
        return _NativeStudentFull.apply(x, emb_all, self, args, *eng.full_params)

    # ---- native LoRA training (native_mode = "train") ---------------------------------------------------------------------
    def native_train_engine(self, forward_only=False):
        """Gradient engine with the LoRA tensors bound (engine_lora.py).  Two instances: one whose tape the backward consumes,
        one for no-grad forwards in between (the distillation step's target forward runs between the student's forward and
        its backward, train_t2v_turbo_v1_lora.py:1022-1190) — separate buffers, same frozen weights."""
        slot = "enc" if forward_only else "grad"
        if getattr(self._engine_box, slot) is None:
            from . import lora
            from .engine_unet_bwd import UNetGradEngine
            make_ops = getattr(self, "_native_ops_factory", None)  # tests substitute the emulated backend
            if make_ops is None:
                from .native import HipOps as make_ops
            eng = UNetGradEngine(self, make_ops())
            eng.bind_lora(lora.lora_parameters(self))
            setattr(self._engine_box, slot, eng)
        return getattr(self._engine_box, slot)

    def _forward_native_train(self, x, timesteps, context, fps, timestep_cond, motion_cond):
        if not (x.is_cuda or getattr(self, "_native_ops_factory", None) is not None):
            raise RuntimeError('native_mode = "train" needs CUDA tensors')
        if context is None:
            raise ValueError("native LoRA training needs the text context")
        grad = torch.is_grad_enabled()
        eng = self.native_train_engine(forward_only=not grad)
        from .nn_util import walk_parameters
        trainable = {id(p) for p in walk_parameters(self) if p.requires_grad}
        if trainable - eng.lora_ids:
            raise RuntimeError('native_mode = "train": only LoRA tensors may require grad (the base weights are frozen packs)')
        if context.requires_grad or (timestep_cond is not None and timestep_cond.requires_grad):
            raise RuntimeError('native_mode = "train": gradients flow to the latents and the LoRA tensors only')
        emb_all = self.conditioning_emb_all(timesteps, fps, timestep_cond, motion_cond)
        args = (timesteps, context, fps, timestep_cond, motion_cond)
        if not grad:
            return eng.forward_tape(x, *args, emb_all=emb_all)
        leaves = [p for mod in eng.engine_leaves() for p in (mod.lora_up.weight, mod.lora_down.weight)]
        return _NativeStudent.apply(x, emb_all, self, args, *leaves)

    def _embedding(self, timesteps, fps, timestep_cond, motion_cond):
        """Time (+ guidance-scale, + motion) and fps embedding, one row per clip (openaimodel3d.py:683-706)."""
        t_emb = sinusoidal_embedding(timesteps, self.model_channels).to(self.dtype)
        cond = self.time_cond_proj(timestep_cond) if timestep_cond is not None else 0.0
        if motion_cond is not None:
            cond = self.combine_proj(torch.cat([cond, self.motion_cond_proj(motion_cond)], dim=1))
        emb = self.time_embed(t_emb + cond)
        if self.fps_cond:
            if type(fps) == int:
                fps = torch.full_like(timesteps, fps)
            emb = emb + self.fps_embedding(sinusoidal_embedding(fps, self.model_channels).to(self.dtype))
        return emb

    def conditioning_emb_all(self, timesteps, fps=16, timestep_cond=None, motion_cond=None):
        """[B, sum of ResBlock widths] fp32: every ResBlock's ``emb_layers(emb)`` side by side, in ``modules()`` order — the
        conditioning input of the native training engine (engine_lora.py).  Differentiable: this M = B-row branch (27
        small leaves) stays with torch autograd, the engine returns d(loss)/d(this tensor)."""
        emb = self._embedding(timesteps, fps, timestep_cond, motion_cond)
        blocks = self.__dict__.get("_resblocks")  # the ResBlock objects never change (LoRA injection swaps leaves inside them)
        if blocks is None:
            blocks = [mod for mod in self.modules() if isinstance(mod, ResBlock)]
            self.__dict__["_resblocks"] = blocks
        return torch.cat([mod.emb_layers(emb) for mod in blocks], dim=1).float()

    def _forward_composite(self, x, timesteps, context, features_adapter, fps, timestep_cond, motion_cond):
        """Reference-semantics torch path (openaimodel3d.py:672-740)."""
        emb = self._embedding(timesteps, fps, timestep_cond, motion_cond)
        b, _, t, hh, ww = x.shape
        context = context.repeat_interleave(repeats=t, dim=0)
        emb = emb.repeat_interleave(repeats=t, dim=0)
        h = x.transpose(1, 2).reshape(b * t, x.shape[1], hh, ww).type(self.dtype)
        hs, adapter_idx = [], 0
        for i, module in enumerate(self.input_blocks):
            h = module(h, emb, context=context, batch_size=b)
            if i == 0 and self.addition_attention:
                h = self.init_attn(h, emb, context=context, batch_size=b)
            if ((i + 1) % 3 == 0) and features_adapter is not None:
                h = h + features_adapter[adapter_idx]
                adapter_idx += 1
            hs.append(h)
        if features_adapter is not None:
            assert len(features_adapter) == adapter_idx, "Wrong features_adapter"
        h = self.middle_block(h, emb, context=context, batch_size=b)
        for module in self.output_blocks:
            h = module(torch.cat([h, hs.pop()], dim=1), emb, context=context, batch_size=b)
        y = self.out(h.type(x.dtype))
        return y.reshape(b, t, y.shape[1], hh, ww).transpose(1, 2)
