"""ModelScope text-to-video denoiser: the ``UNet3DConditionModel`` that the
reference's ``model_scope/unet_3d_condition.py:86-503`` / ``model_scope/unet_3d_blocks.py:268-875`` assemble from
diffusers 0.30 leaf classes (``ResnetBlock2D``, ``TemporalConvLayer``, ``Transformer2DModel``,
``TransformerTemporalModel``, ``Downsample2D``, ``Upsample2D``, ``TimestepEmbedding`` — pinned only in the reference's
``cog.yaml:14-15``; diffusers is NOT vendored under /root/reference and not installed here).

Parameter names / shapes follow the diffusers state-dict layout (``down_blocks.0.resnets.0.norm1.weight``,
``...attentions.0.transformer_blocks.0.attn1.to_q.weight``, ``...temp_convs.0.conv1.0.weight``, ``time_embedding.cond_proj``)
so a ModelScope / t2v-turbo-MS checkpoint loads with ``load_state_dict``.  The leaf arithmetic is the same vocabulary
as the VideoCrafter2 path (GroupNorm -> SiLU -> conv, (3,1,1) temporal convs, pre-LN transformer blocks with GEGLU,
per-pixel temporal self-attention), so the leaves ARE the VC2 leaf classes of ``unet3d.py`` under their diffusers
attribute names, and the native path reuses every HIP kernel of the VC2 engine (``engine_ms.MSUNetEngine``).

PARITY UNPINNED: the reference ships no test or golden vector for this backbone and its leaf classes live in an absent
third-party package; the semantics below are the published diffusers 0.30.0 ones restated, cross-checked only against
``oracle/ms_unet_oracle.py`` (an independent functional restatement from the same published description)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .nn_util import EngineBox, sinusoidal_embedding
from .unet3d import SpatialTransformer, TemporalConvBlock, TemporalTransformer


# =================================================================================== leaves
class ResnetBlock2D(nn.Module):
    """diffusers ``ResnetBlock2D`` (time_embedding_norm="default", pre_norm, output_scale_factor 1):
    conv1(silu(norm1 x)) + time_emb_proj(silu temb) -> conv2(drop(silu(norm2 .))) ; + (1x1 conv_shortcut)(x)."""

    def __init__(self, in_channels, out_channels, temb_channels, eps=1e-5, groups=32, dropout=0.0, output_scale_factor=1.0):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.output_scale_factor = output_scale_factor
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb):
        h = self.conv1(self.nonlinearity(self.norm1(x)))
        h = h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return (x + h) / self.output_scale_factor


class TemporalConvLayer(TemporalConvBlock):
    """diffusers ``TemporalConvLayer``: 4 x [GroupNorm(32) -> SiLU -> (Dropout) -> Conv3d (3,1,1)] + identity on the
    (b c f h w) view of a (b f) c h w tensor; same structure and key names as VC2's TemporalConvBlock."""

    def __init__(self, in_dim, out_dim=None, dropout=0.0):
        super().__init__(in_dim, out_dim, dropout=dropout)

    def forward(self, x, num_frames=1):  # x: (b f) c h w
        n, c, h, w = x.shape
        x5 = x.reshape(n // num_frames, num_frames, c, h, w).transpose(1, 2)
        y = super().forward(x5)
        return y.transpose(1, 2).reshape(n, c, h, w)


class Transformer2DModel(SpatialTransformer):
    """diffusers ``Transformer2DModel`` with use_linear_projection=True, one BasicTransformerBlock (self-attn, text
    cross-attn, GEGLU FF), GroupNorm eps 1e-6: identical to VC2's SpatialTransformer(use_linear=True)."""

    def __init__(self, num_attention_heads, attention_head_dim, in_channels, num_layers=1, cross_attention_dim=None):
        super().__init__(in_channels, num_attention_heads, attention_head_dim, depth=num_layers,
                         context_dim=cross_attention_dim, use_checkpoint=False, use_linear=True)

    def forward(self, x, encoder_hidden_states=None):
        return super().forward(x, context=encoder_hidden_states)


class TransformerTemporalModel(TemporalTransformer):
    """diffusers ``TransformerTemporalModel`` (double_self_attention: both attentions are temporal self-attention):
    identical to VC2's TemporalTransformer(use_linear=True, only_self_att=True)."""

    def __init__(self, num_attention_heads, attention_head_dim, in_channels, num_layers=1):
        super().__init__(in_channels, num_attention_heads, attention_head_dim, depth=num_layers, use_checkpoint=False,
                         use_linear=True, only_self_att=True)

    def forward(self, x, num_frames=1):  # x: (b f) c h w
        n, c, h, w = x.shape
        x5 = x.reshape(n // num_frames, num_frames, c, h, w).transpose(1, 2)
        return super().forward(x5).transpose(1, 2).reshape(n, c, h, w)


class Downsample2D(nn.Module):
    def __init__(self, channels, out_channels=None, padding=1):
        super().__init__()
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=padding)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, padding=1)

    def forward(self, x, output_size=None):
        if output_size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=output_size, mode="nearest")
        return self.conv(x)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, cond_proj_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.cond_proj = nn.Linear(cond_proj_dim, in_channels, bias=False) if cond_proj_dim is not None else None
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample, condition=None):
        if condition is not None:
            sample = sample + self.cond_proj(condition)
        return self.linear_2(self.act(self.linear_1(sample)))


# =================================================================================== blocks (unet_3d_blocks.py)
class _Block3D(nn.Module):
    has_cross_attention = False

    def _layer(self, i, h, temb, ctx, num_frames):
        """resnet -> temporal conv -> [spatial transformer -> temporal transformer] (unet_3d_blocks.py:547-561)."""
        h = self.resnets[i](h, temb)
        if num_frames > 1:
            h = self.temp_convs[i](h, num_frames=num_frames)
        if self.has_cross_attention:
            h = self.attentions[i](h, encoder_hidden_states=ctx)
            if num_frames > 1:
                h = self.temp_attentions[i](h, num_frames=num_frames)
        return h


class CrossAttnDownBlock3D(_Block3D):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, resnet_eps, resnet_groups, head_dim,
                 cross_attention_dim, add_downsample, downsample_padding=1, with_attention=True):
        super().__init__()
        self.has_cross_attention = with_attention
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb_channels,
                                                    eps=resnet_eps, groups=resnet_groups) for i in range(num_layers)])
        self.temp_convs = nn.ModuleList([TemporalConvLayer(out_channels, out_channels, dropout=0.1) for _ in range(num_layers)])
        if with_attention:
            self.attentions = nn.ModuleList([Transformer2DModel(out_channels // head_dim, head_dim, out_channels, 1,
                                                                cross_attention_dim) for _ in range(num_layers)])
            self.temp_attentions = nn.ModuleList([TransformerTemporalModel(out_channels // head_dim, head_dim, out_channels, 1)
                                                  for _ in range(num_layers)])
        self.downsamplers = (nn.ModuleList([Downsample2D(out_channels, out_channels, downsample_padding)])
                             if add_downsample else None)

    def forward(self, h, temb=None, encoder_hidden_states=None, num_frames=1):
        outs = ()
        for i in range(len(self.resnets)):
            h = self._layer(i, h, temb, encoder_hidden_states, num_frames)
            outs += (h,)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            outs += (h,)
        return h, outs


class DownBlock3D(CrossAttnDownBlock3D):
    def __init__(self, in_channels, out_channels, temb_channels, num_layers, resnet_eps, resnet_groups, add_downsample,
                 downsample_padding=1):
        super().__init__(in_channels, out_channels, temb_channels, num_layers, resnet_eps, resnet_groups, 64, None,
                         add_downsample, downsample_padding, with_attention=False)


class UNetMidBlock3DCrossAttn(nn.Module):
    """resnet, temp_conv, then [attn, temp_attn, resnet, temp_conv] (unet_3d_blocks.py:268-420)."""

    def __init__(self, in_channels, temb_channels, resnet_eps, resnet_groups, head_dim, cross_attention_dim, num_layers=1):
        super().__init__()
        n = num_layers + 1
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels, in_channels, temb_channels, eps=resnet_eps,
                                                    groups=resnet_groups) for _ in range(n)])
        self.temp_convs = nn.ModuleList([TemporalConvLayer(in_channels, in_channels, dropout=0.1) for _ in range(n)])
        self.attentions = nn.ModuleList([Transformer2DModel(in_channels // head_dim, head_dim, in_channels, 1,
                                                            cross_attention_dim) for _ in range(num_layers)])
        self.temp_attentions = nn.ModuleList([TransformerTemporalModel(in_channels // head_dim, head_dim, in_channels, 1)
                                              for _ in range(num_layers)])

    def forward(self, h, temb=None, encoder_hidden_states=None, num_frames=1):
        h = self.resnets[0](h, temb)
        h = self.temp_convs[0](h, num_frames=num_frames)  # the reference applies this one even for a single frame
        for attn, tattn, resnet, tconv in zip(self.attentions, self.temp_attentions, self.resnets[1:], self.temp_convs[1:]):
            h = attn(h, encoder_hidden_states=encoder_hidden_states)
            if num_frames > 1:
                h = tattn(h, num_frames=num_frames)
            h = resnet(h, temb)
            if num_frames > 1:
                h = tconv(h, num_frames=num_frames)
        return h


class CrossAttnUpBlock3D(_Block3D):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers, resnet_eps, resnet_groups,
                 head_dim, cross_attention_dim, add_upsample, with_attention=True):
        super().__init__()
        self.has_cross_attention = with_attention
        resnets = []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            cin = prev_output_channel if i == 0 else out_channels
            resnets.append(ResnetBlock2D(cin + skip, out_channels, temb_channels, eps=resnet_eps, groups=resnet_groups))
        self.resnets = nn.ModuleList(resnets)
        self.temp_convs = nn.ModuleList([TemporalConvLayer(out_channels, out_channels, dropout=0.1) for _ in range(num_layers)])
        if with_attention:
            self.attentions = nn.ModuleList([Transformer2DModel(out_channels // head_dim, head_dim, out_channels, 1,
                                                                cross_attention_dim) for _ in range(num_layers)])
            self.temp_attentions = nn.ModuleList([TransformerTemporalModel(out_channels // head_dim, head_dim, out_channels, 1)
                                                  for _ in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, out_channels)]) if add_upsample else None

    def forward(self, h, res_hidden_states_tuple, temb=None, encoder_hidden_states=None, upsample_size=None, num_frames=1):
        for i in range(len(self.resnets)):
            skip = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            h = self._layer(i, torch.cat([h, skip], dim=1), temb, encoder_hidden_states, num_frames)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h, upsample_size)
        return h


class UpBlock3D(CrossAttnUpBlock3D):
    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers, resnet_eps, resnet_groups,
                 add_upsample):
        super().__init__(in_channels, out_channels, prev_output_channel, temb_channels, num_layers, resnet_eps, resnet_groups,
                         64, None, add_upsample, with_attention=False)


class UNet3DConditionOutput:
    def __init__(self, sample):
        self.sample = sample


# =================================================================================== the denoiser
class UNet3DConditionModel(nn.Module):
    """Constructor keywords and ``forward`` as ``model_scope/unet_3d_condition.py:86-107,329-503`` (attention masks,
    class labels, ControlNet residuals and attention slicing are not used by t2v-turbo and not supported)."""

    def __init__(self, sample_size=None, in_channels=4, out_channels=4,
                 down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
                 up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
                 block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1,
                 mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5, time_cond_proj_dim=None,
                 cross_attention_dim=1024, attention_head_dim=64):
        super().__init__()
        if len(down_block_types) != len(up_block_types):
            raise ValueError(f"Must provide the same number of `down_block_types` as `up_block_types`. "
                             f"`down_block_types`: {down_block_types}. `up_block_types`: {up_block_types}.")
        if len(block_out_channels) != len(down_block_types):
            raise ValueError(f"Must provide the same number of `block_out_channels` as `down_block_types`. "
                             f"`block_out_channels`: {block_out_channels}. `down_block_types`: {down_block_types}.")
        if not isinstance(attention_head_dim, int) and len(attention_head_dim) != len(down_block_types):
            raise ValueError("Must provide the same number of `attention_head_dim` as `down_block_types`.")
        if act_fn != "silu" or mid_block_scale_factor != 1 or norm_num_groups is None:
            raise NotImplementedError("UNet3DConditionModel: option outside the ModelScope / t2v-turbo-MS configuration")
        self.config = dict(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                           down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
                           block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                           downsample_padding=downsample_padding, norm_num_groups=norm_num_groups, norm_eps=norm_eps,
                           time_cond_proj_dim=time_cond_proj_dim, cross_attention_dim=cross_attention_dim,
                           attention_head_dim=attention_head_dim)
        self.sample_size = sample_size
        self.in_channels, self.out_channels = in_channels, out_channels
        self.model_channels = block_out_channels[0]
        self.time_cond_proj_dim = time_cond_proj_dim
        self.fps_cond = False  # engine plumbing shared with the VC2 UNet
        ch0 = block_out_channels[0]
        self.conv_in = nn.Conv2d(in_channels, ch0, 3, padding=1)
        ted = ch0 * 4
        self.time_embedding = TimestepEmbedding(ch0, ted, cond_proj_dim=time_cond_proj_dim)
        hd = attention_head_dim
        self.transformer_in = TransformerTemporalModel(8, hd if isinstance(hd, int) else hd[0], ch0, 1)
        hds = (hd,) * len(down_block_types) if isinstance(hd, int) else tuple(hd)
        self.down_blocks = nn.ModuleList()
        out_ch = ch0
        for i, kind in enumerate(down_block_types):
            in_ch, out_ch = out_ch, block_out_channels[i]
            final = i == len(block_out_channels) - 1
            if kind == "CrossAttnDownBlock3D":
                blk = CrossAttnDownBlock3D(in_ch, out_ch, ted, layers_per_block, norm_eps, norm_num_groups, hds[i],
                                           cross_attention_dim, not final, downsample_padding)
            elif kind == "DownBlock3D":
                blk = DownBlock3D(in_ch, out_ch, ted, layers_per_block, norm_eps, norm_num_groups, not final, downsample_padding)
            else:
                raise ValueError(f"{kind} does not exist.")
            self.down_blocks.append(blk)
        self.mid_block = UNetMidBlock3DCrossAttn(block_out_channels[-1], ted, norm_eps, norm_num_groups, hds[-1],
                                                 cross_attention_dim)
        self.num_upsamplers = 0
        self.up_blocks = nn.ModuleList()
        rev, rev_hd = list(reversed(block_out_channels)), list(reversed(hds))
        out_ch = rev[0]
        for i, kind in enumerate(up_block_types):
            final = i == len(block_out_channels) - 1
            prev, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(block_out_channels) - 1)]
            if not final:
                self.num_upsamplers += 1
            if kind == "CrossAttnUpBlock3D":
                blk = CrossAttnUpBlock3D(in_ch, out_ch, prev, ted, layers_per_block + 1, norm_eps, norm_num_groups, rev_hd[i],
                                         cross_attention_dim, not final)
            elif kind == "UpBlock3D":
                blk = UpBlock3D(in_ch, out_ch, prev, ted, layers_per_block + 1, norm_eps, norm_num_groups, not final)
            else:
                raise ValueError(f"{kind} does not exist.")
            self.up_blocks.append(blk)
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, ch0, eps=norm_eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(ch0, out_channels, 3, padding=1)
        self._engine_box = EngineBox()

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, timestep_cond=None, attention_mask=None,
                cross_attention_kwargs=None, down_block_additional_residuals=None, mid_block_additional_residual=None,
                return_dict=True):
        if (class_labels is not None or attention_mask is not None or cross_attention_kwargs is not None
                or down_block_additional_residuals is not None or mid_block_additional_residual is not None):
            raise NotImplementedError("class labels / attention masks / ControlNet residuals are not used by t2v-turbo")
        timesteps = timestep
        if not torch.is_tensor(timesteps):
            timesteps = torch.tensor([timesteps], dtype=torch.float64 if isinstance(timestep, float) else torch.int64,
                                     device=sample.device)
        elif timesteps.dim() == 0:
            timesteps = timesteps[None].to(sample.device)
        timesteps = timesteps.expand(sample.shape[0])
        native = getattr(self, "native_mode", "auto") != "off"
        if native and sample.is_cuda and sample.shape[2] > 1 and not (torch.is_grad_enabled() and self._needs_grad(
                sample, encoder_hidden_states, timestep_cond)):
            if any(s % (2 ** self.num_upsamplers) for s in sample.shape[-2:]):
                raise NotImplementedError("native ModelScope path needs H, W divisible by 2**num_upsamplers")
            out = self.native_engine()(sample, timesteps.contiguous(), encoder_hidden_states, 16, timestep_cond, None)
        else:
            out = self._forward_composite(sample, timesteps, encoder_hidden_states, timestep_cond)
        return UNet3DConditionOutput(out) if return_dict else (out,)

    def _needs_grad(self, *tensors):
        if any(t is not None and t.requires_grad for t in tensors):
            return True
        return any(p.requires_grad for p in self.parameters())

    def native_engine(self):
        if self._engine_box.engine is None:
            from .engine_ms import MSUNetEngine
            from .native import HipOps
            self._engine_box.engine = MSUNetEngine(self, HipOps())
        return self._engine_box.engine

    def _forward_composite(self, sample, timesteps, ctx, timestep_cond):
        """Reference-semantics torch path (unet_3d_condition.py:360-503)."""
        up_factor = 2 ** self.num_upsamplers
        forward_upsample_size = any(s % up_factor != 0 for s in sample.shape[-2:])
        b, _, nf, hh, ww = sample.shape
        t_emb = sinusoidal_embedding(timesteps, self.model_channels).to(self.dtype)  # Timesteps(ch0, flip_sin_to_cos, 0)
        emb = self.time_embedding(t_emb, timestep_cond)
        emb = emb.repeat_interleave(repeats=nf, dim=0)
        ctx = ctx.repeat_interleave(repeats=nf, dim=0)
        h = sample.permute(0, 2, 1, 3, 4).reshape(b * nf, -1, hh, ww).to(self.dtype)
        h = self.conv_in(h)
        if nf > 1:
            h = self.transformer_in(h, num_frames=nf)
        res = (h,)
        for blk in self.down_blocks:
            h, outs = blk(h, temb=emb, encoder_hidden_states=ctx, num_frames=nf)
            res += outs
        h = self.mid_block(h, emb, encoder_hidden_states=ctx, num_frames=nf)
        upsample_size = None
        for i, blk in enumerate(self.up_blocks):
            n = len(blk.resnets)
            skips, res = res[-n:], res[:-n]
            if i != len(self.up_blocks) - 1 and forward_upsample_size:
                upsample_size = res[-1].shape[2:]
            h = blk(h, skips, temb=emb, encoder_hidden_states=ctx, upsample_size=upsample_size, num_frames=nf)
        h = self.conv_out(self.conv_act(self.conv_norm_out(h)))
        return h[None, :].reshape((-1, nf) + h.shape[1:]).permute(0, 2, 1, 3, 4)
