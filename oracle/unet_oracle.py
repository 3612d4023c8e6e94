"""fp32 CPU restatement of the VideoCrafter2 3D-UNet forward (oracle, test-only).

Functional: takes a flat ``state_dict`` (reference key names) + the UNet kwargs
of ``configs/inference_t2v_512_v2.0.yaml`` and reproduces
``UNetModel.forward`` (reference ``lvdm/modules/networks/openaimodel3d.py:672-740``)
with plain torch ops.  Pinned against the imported reference by
``tests/golden/make_golden.py`` -> ``tests/test_oracle_golden.py``.
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------
# architecture walk (restates the ctor loops, openaimodel3d.py:432-670)
# ---------------------------------------------------------------------------
def unet_layout(cfg):
    """Return the per-block child kinds, mirroring the reference ctor.

    Each entry: (prefix, [(kind, child_index, info), ...]).
    kinds: 'conv_in', 'res', 'spatial', 'temporal', 'down', 'up'.
    """
    mc = cfg["model_channels"]
    mult = list(cfg["channel_mult"])
    nrb = cfg["num_res_blocks"]
    attn_res = set(cfg["attention_resolutions"])
    nhc = cfg.get("num_head_channels", -1)
    nh = cfg.get("num_heads", -1)
    temporal_attention = cfg.get("temporal_attention", True)

    def heads_of(ch):
        if nhc == -1:
            return nh, ch // nh
        return ch // nhc, nhc

    inp = [("input_blocks.0", [("conv_in", 0, dict(cin=cfg["in_channels"], cout=mc))])]
    chans = [mc]
    ch, ds = mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            kids = [("res", 0, dict(cin=ch, cout=m * mc))]
            ch = m * mc
            if ds in attn_res:
                h, d = heads_of(ch)
                kids.append(("spatial", 1, dict(ch=ch, heads=h, dh=d)))
                if temporal_attention:
                    kids.append(("temporal", 2, dict(ch=ch, heads=h, dh=d)))
            inp.append((f"input_blocks.{len(inp)}", kids))
            chans.append(ch)
        if level != len(mult) - 1:
            inp.append((f"input_blocks.{len(inp)}", [("down", 0, dict(ch=ch))]))
            chans.append(ch)
            ds *= 2
    h, d = heads_of(ch)
    mid = [("res", 0, dict(cin=ch, cout=ch)), ("spatial", 1, dict(ch=ch, heads=h, dh=d))]
    if temporal_attention:
        mid.append(("temporal", 2, dict(ch=ch, heads=h, dh=d)))
    mid.append(("res", len(mid), dict(cin=ch, cout=ch)))
    out = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            kids = [("res", 0, dict(cin=ch + ich, cout=m * mc))]
            ch = m * mc
            if ds in attn_res:
                h, d = heads_of(ch)
                kids.append(("spatial", len(kids), dict(ch=ch, heads=h, dh=d)))
                if temporal_attention:
                    kids.append(("temporal", len(kids), dict(ch=ch, heads=h, dh=d)))
            if level and i == nrb:
                kids.append(("up", len(kids), dict(ch=ch)))
                ds //= 2
            out.append((f"output_blocks.{len(out)}", kids))
    return inp, mid, out


# ---------------------------------------------------------------------------
# primitives
# ---------------------------------------------------------------------------
def timestep_embedding(t, dim, max_period=10000):
    """cos||sin sinusoid (lvdm/models/utils_diffusion.py:8-32)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _gn(sd, p, x, eps):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def attention(sd, p, x, context, heads):
    """CrossAttention.forward vanilla path (lvdm/modules/attention.py:102-164)."""
    ctx = x if context is None else context
    q = F.linear(x, sd[p + ".to_q.weight"])
    k = F.linear(ctx, sd[p + ".to_k.weight"])
    v = F.linear(ctx, sd[p + ".to_v.weight"])
    b, n, inner = q.shape
    dh = inner // heads

    def split(t):
        return t.reshape(t.shape[0], t.shape[1], heads, dh).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    sim = torch.matmul(q, k.transpose(-1, -2)) * dh ** -0.5
    probs = sim.softmax(dim=-1)
    o = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(b, n, inner)
    return _lin(sd, p + ".to_out.0", o), probs


def transformer_block(sd, p, x, context, heads, self_only):
    """BasicTransformerBlock._forward (attention.py:300-311); attn2 is a self
    attention when the block was built with context_dim=None (temporal,
    attention.py:446-447)."""
    a1, probs = attention(sd, p + ".attn1", _ln(sd, p + ".norm1", x), None, heads)
    x = a1 + x
    a2, _ = attention(sd, p + ".attn2", _ln(sd, p + ".norm2", x), None if self_only else context, heads)
    x = a2 + x
    h = _lin(sd, p + ".ff.net.0.proj", _ln(sd, p + ".norm3", x))
    a, gate = h.chunk(2, dim=-1)  # GEGLU attention.py:521-523
    x = _lin(sd, p + ".ff.net.2", a * F.gelu(gate)) + x
    return x, probs


def spatial_transformer(sd, p, x, context, info):
    """SpatialTransformer.forward, use_linear=True (attention.py:373-389)."""
    n, c, h, w = x.shape
    y = _gn(sd, p + ".norm", x, 1e-6)
    y = y.permute(0, 2, 3, 1).reshape(n, h * w, c)
    proj_in_w = sd[p + ".proj_in.weight"]
    y = F.linear(y, proj_in_w.reshape(proj_in_w.shape[0], -1), sd[p + ".proj_in.bias"])
    y, _ = transformer_block(sd, p + ".transformer_blocks.0", y, context, info["heads"], False)
    proj_out_w = sd[p + ".proj_out.weight"]
    y = F.linear(y, proj_out_w.reshape(proj_out_w.shape[0], -1), sd[p + ".proj_out.bias"])
    y = y.reshape(n, h, w, c).permute(0, 3, 1, 2)
    return y + x


def temporal_transformer(sd, p, x5, heads, probs_out=None):
    """TemporalTransformer.forward, only_self_att=True (attention.py:471-513).
    ``proj_in``/``proj_out`` are Linear (use_linear) or Conv1d k=1 (init_attn,
    openaimodel3d.py:439-453): both are a per-token matmul."""
    b, c, t, h, w = x5.shape
    y = _gn(sd, p + ".norm", x5, 1e-6)
    y = y.permute(0, 3, 4, 2, 1).reshape(b * h * w, t, c)
    wi = sd[p + ".proj_in.weight"]
    y = F.linear(y, wi.reshape(wi.shape[0], -1), sd[p + ".proj_in.bias"])
    y, probs = transformer_block(sd, p + ".transformer_blocks.0", y, None, heads, True)
    if probs_out is not None:
        probs_out[p + ".transformer_blocks.0.attn1"] = probs.reshape(-1, t, t)
    wo = sd[p + ".proj_out.weight"]
    y = F.linear(y, wo.reshape(wo.shape[0], -1), sd[p + ".proj_out.bias"])
    y = y.reshape(b, h, w, t, c).permute(0, 4, 3, 1, 2)
    return y + x5


def temporal_conv_block(sd, p, x5):
    """TemporalConvBlock.forward, eval mode (openaimodel3d.py:302-309)."""
    y = x5
    for i, conv_idx in ((1, 2), (2, 3), (3, 3), (4, 3)):
        y = F.silu(_gn(sd, f"{p}.conv{i}.0", y, 1e-5))
        y = F.conv3d(y, sd[f"{p}.conv{i}.{conv_idx}.weight"], sd[f"{p}.conv{i}.{conv_idx}.bias"], padding=(1, 0, 0))
    return y + x5


def res_block(sd, p, x, emb, b, temporal_conv):
    """ResBlock._forward, no up/down, no scale-shift (openaimodel3d.py:223-254)."""
    h = F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5))
    h = F.conv2d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    e = _lin(sd, p + ".emb_layers.1", F.silu(emb))
    h = h + e[:, :, None, None]
    h = F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5))
    h = F.conv2d(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if p + ".skip_connection.weight" in sd:
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    h = x + h
    if temporal_conv:
        n, c, hh, ww = h.shape
        h5 = h.reshape(b, n // b, c, hh, ww).permute(0, 2, 1, 3, 4)
        h5 = temporal_conv_block(sd, p + ".temopral_conv", h5)
        h = h5.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)
    return h


def _run_block(sd, prefix, kids, h, emb, context, b, cfg, probs_out):
    for kind, idx, info in kids:
        p = f"{prefix}.{idx}"
        if kind == "conv_in":
            h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
        elif kind == "res":
            h = res_block(sd, p, h, emb, b, cfg.get("temporal_conv", False))
        elif kind == "spatial":
            h = spatial_transformer(sd, p, h, context, info)
        elif kind == "temporal":
            n, c, hh, ww = h.shape
            h5 = h.reshape(b, n // b, c, hh, ww).permute(0, 2, 1, 3, 4)
            h5 = temporal_transformer(sd, p, h5, info["heads"], probs_out)
            h = h5.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)
        elif kind == "down":  # Downsample conv s2 (openaimodel3d.py:63-79)
            h = F.conv2d(h, sd[p + ".op.weight"], sd[p + ".op.bias"], stride=2, padding=1)
        elif kind == "up":  # Upsample nearest x2 + conv (openaimodel3d.py:102-112)
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = F.conv2d(h, sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)
        else:
            raise ValueError(kind)
    return h


@torch.no_grad()
def unet_forward(sd, cfg, x, timesteps, context, fps=16, timestep_cond=None, motion_cond=None,
                 probs_out=None):
    """UNetModel.forward (openaimodel3d.py:672-740), fp32, eval mode."""
    if any(v.dtype != torch.float32 for v in sd.values()):
        sd = {k: v.float() for k, v in sd.items()}
    mc = cfg["model_channels"]
    x = x.float()
    context = context.float()
    t_emb = timestep_embedding(timesteps, mc)
    if timestep_cond is not None:
        cond = F.linear(timestep_cond.float(), sd["time_cond_proj.weight"])
    else:
        cond = 0.0
    if motion_cond is not None:
        assert timestep_cond is not None
        m = F.linear(motion_cond.float(), sd["motion_cond_proj.weight"])
        cond = F.linear(torch.cat([cond, m], dim=1), sd["combine_proj.weight"])
    emb = _lin(sd, "time_embed.2", F.silu(_lin(sd, "time_embed.0", t_emb + cond)))
    if cfg.get("fps_cond", False):
        if isinstance(fps, int):
            fps = torch.full_like(timesteps, fps)
        f_emb = timestep_embedding(fps, mc)
        emb = emb + _lin(sd, "fps_embedding.2", F.silu(_lin(sd, "fps_embedding.0", f_emb)))
    b, _, t, hh, ww = x.shape
    context = context.repeat_interleave(t, dim=0)
    emb = emb.repeat_interleave(t, dim=0)
    h = x.permute(0, 2, 1, 3, 4).reshape(b * t, x.shape[1], hh, ww)

    inp, mid, out = unet_layout(cfg)
    hs = []
    for i, (prefix, kids) in enumerate(inp):
        h = _run_block(sd, prefix, kids, h, emb, context, b, cfg, probs_out)
        if i == 0 and cfg.get("addition_attention", False):
            n, c, h2, w2 = h.shape
            h5 = h.reshape(b, t, c, h2, w2).permute(0, 2, 1, 3, 4)
            h5 = temporal_transformer(sd, "init_attn.0", h5, 8, probs_out)
            h = h5.permute(0, 2, 1, 3, 4).reshape(n, c, h2, w2)
        hs.append(h)
    h = _run_block(sd, "middle_block", mid, h, emb, context, b, cfg, probs_out)
    for prefix, kids in out:
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_block(sd, prefix, kids, h, emb, context, b, cfg, probs_out)
    y = F.silu(_gn(sd, "out.0", h, 1e-5))
    y = F.conv2d(y, sd["out.2.weight"], sd["out.2.bias"], padding=1)
    return y.reshape(b, t, y.shape[1], hh, ww).permute(0, 2, 1, 3, 4).contiguous()
