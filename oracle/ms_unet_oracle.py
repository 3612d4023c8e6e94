"""fp32 CPU restatement of the ModelScope denoiser forward (oracle, test-only) — PARITY UNPINNED.

``UNet3DConditionModel.forward`` as wired by the reference's ``model_scope/unet_3d_condition.py:329-503`` and
``model_scope/unet_3d_blocks.py:268-875``.  The leaf classes those files instantiate come from **diffusers**
(pinned ``diffusers==0.30.0`` only in the reference's ``cog.yaml:14-15``; not vendored, not installed here), so their
arithmetic is restated from the published 0.30.0 semantics:
  ResnetBlock2D          GroupNorm -> SiLU -> conv3x3 ; + Linear(SiLU(temb)) ; GroupNorm -> SiLU -> conv3x3 ; + 1x1 shortcut
  TemporalConvLayer      4 x [GroupNorm(32, eps 1e-5) -> SiLU -> Conv3d (3,1,1)] + identity on the (b c f h w) view
  Transformer2DModel     GroupNorm(eps 1e-6) -> Linear proj_in -> pre-LN block (self-attn, text cross-attn, GEGLU) -> proj_out -> +x
  TransformerTemporalModel   the same over the frame axis of each pixel, both attentions self (double_self_attention)
  Timesteps(flip_sin_to_cos=True, shift 0) = cos||sin ; TimestepEmbedding = linear_2(SiLU(linear_1(t_emb + cond_proj(w))))
The reference has no test, golden vector or fixture for this backbone: this oracle is pinned to nothing but the text
above, and parity claims made with it are "partial" by construction.  State-dict keys are the diffusers ones."""
import torch
import torch.nn.functional as F

from .unet_oracle import _gn, _lin, spatial_transformer, temporal_conv_block, temporal_transformer, timestep_embedding


def resnet_block(sd, p, x, temb, eps):
    h = F.conv2d(F.silu(_gn(sd, p + ".norm1", x, eps)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = h + _lin(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = F.conv2d(F.silu(_gn(sd, p + ".norm2", h, eps)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if p + ".conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def _to5(x, nf):
    n, c, h, w = x.shape
    return x.reshape(n // nf, nf, c, h, w).permute(0, 2, 1, 3, 4)


def _to4(x5):
    b, c, f, h, w = x5.shape
    return x5.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)


def temp_conv(sd, p, x, nf):
    return _to4(temporal_conv_block(sd, p, _to5(x, nf)))


def temp_attn(sd, p, x, nf, heads):
    return _to4(temporal_transformer(sd, p, _to5(x, nf), heads))


def _layer(sd, p, i, h, temb, ctx, nf, eps, attn):
    """resnet -> temp_conv -> [attn -> temp_attn] (unet_3d_blocks.py:547-561)."""
    h = resnet_block(sd, f"{p}.resnets.{i}", h, temb, eps)
    h = temp_conv(sd, f"{p}.temp_convs.{i}", h, nf)
    if attn:
        heads = h.shape[1] // 64
        h = spatial_transformer(sd, f"{p}.attentions.{i}", h, ctx, dict(heads=heads))
        h = temp_attn(sd, f"{p}.temp_attentions.{i}", h, nf, heads)
    return h


@torch.no_grad()
def ms_unet_forward(sd, cfg, sample, timesteps, ctx, timestep_cond=None):
    """sample (b, c, f, h, w) -> (b, c, f, h, w) (unet_3d_condition.py:329-503), attention_head_dim == 64."""
    sd = {k: v.float() for k, v in sd.items()}
    eps = cfg.get("norm_eps", 1e-5)
    chans = list(cfg.get("block_out_channels", (320, 640, 1280, 1280)))
    down_types = list(cfg.get("down_block_types", ("CrossAttnDownBlock3D",) * 3 + ("DownBlock3D",)))
    up_types = list(cfg.get("up_block_types", ("UpBlock3D",) + ("CrossAttnUpBlock3D",) * 3))
    lpb = cfg.get("layers_per_block", 2)
    b, _, nf, hh, ww = sample.shape
    t_emb = timestep_embedding(timesteps.expand(b), chans[0])
    if timestep_cond is not None:
        t_emb = t_emb + F.linear(timestep_cond.float(), sd["time_embedding.cond_proj.weight"])
    emb = _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", t_emb)))
    emb = emb.repeat_interleave(nf, dim=0)
    ctx = ctx.float().repeat_interleave(nf, dim=0)
    h = sample.float().permute(0, 2, 1, 3, 4).reshape(b * nf, -1, hh, ww)
    h = F.conv2d(h, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    h = temp_attn(sd, "transformer_in", h, nf, 8)
    res = [h]
    for i, kind in enumerate(down_types):
        p = f"down_blocks.{i}"
        for j in range(lpb):
            h = _layer(sd, p, j, h, emb, ctx, nf, eps, kind == "CrossAttnDownBlock3D")
            res.append(h)
        if i != len(chans) - 1:
            h = F.conv2d(h, sd[p + ".downsamplers.0.conv.weight"], sd[p + ".downsamplers.0.conv.bias"], stride=2,
                         padding=cfg.get("downsample_padding", 1))
            res.append(h)
    h = resnet_block(sd, "mid_block.resnets.0", h, emb, eps)
    h = temp_conv(sd, "mid_block.temp_convs.0", h, nf)
    heads = h.shape[1] // 64
    h = spatial_transformer(sd, "mid_block.attentions.0", h, ctx, dict(heads=heads))
    h = temp_attn(sd, "mid_block.temp_attentions.0", h, nf, heads)
    h = resnet_block(sd, "mid_block.resnets.1", h, emb, eps)
    h = temp_conv(sd, "mid_block.temp_convs.1", h, nf)
    for i, kind in enumerate(up_types):
        p = f"up_blocks.{i}"
        for j in range(lpb + 1):
            h = torch.cat([h, res.pop()], dim=1)
            h = _layer(sd, p, j, h, emb, ctx, nf, eps, kind == "CrossAttnUpBlock3D")
        if i != len(chans) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[p + ".upsamplers.0.conv.weight"], sd[p + ".upsamplers.0.conv.bias"], padding=1)
    h = F.silu(_gn(sd, "conv_norm_out", h, eps))
    h = F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
    return h.reshape(b, nf, -1, hh, ww).permute(0, 2, 1, 3, 4)
