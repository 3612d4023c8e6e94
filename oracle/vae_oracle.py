"""fp32 CPU restatement of the KL-VAE decode path (oracle, test-only).

``decode_first_stage_2DAE`` (reference ``lvdm/models/ddpm3d.py:666-679``) ->
``AutoencoderKL.decode`` (``lvdm/models/autoencoder.py:110-113``) ->
``Decoder.forward`` (``lvdm/modules/networks/ae_modules.py:602-641``).
State-dict keys are those of ``AutoencoderKL`` (``post_quant_conv.*``,
``decoder.*``).
"""
import torch
import torch.nn.functional as F


def _gn(sd, p, x):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], 1e-6)  # Normalize ae_modules.py:16-19


def _swish(x):
    return x * torch.sigmoid(x)  # ae_modules.py:11-13


def _conv(sd, p, x, padding):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=padding)


def resnet_block(sd, p, x):
    """ResnetBlock.forward with temb=None (ae_modules.py:183-203)."""
    h = _conv(sd, p + ".conv1", _swish(_gn(sd, p + ".norm1", x)), 1)
    h = _conv(sd, p + ".conv2", _swish(_gn(sd, p + ".norm2", h)), 1)
    if p + ".nin_shortcut.weight" in sd:
        x = _conv(sd, p + ".nin_shortcut", x, 0)
    elif p + ".conv_shortcut.weight" in sd:
        x = _conv(sd, p + ".conv_shortcut", x, 1)
    return x + h


def attn_block(sd, p, x):
    """AttnBlock.forward: single-head attention over h*w (ae_modules.py:48-73)."""
    h_ = _gn(sd, p + ".norm", x)
    q = _conv(sd, p + ".q", h_, 0)
    k = _conv(sd, p + ".k", h_, 0)
    v = _conv(sd, p + ".v", h_, 0)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.bmm(q, k) * (int(c) ** -0.5)
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, hh * ww)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, p + ".proj_out", h_, 0)


@torch.no_grad()
def decoder_forward(sd, ddconfig, z, prefix="decoder"):
    """Decoder.forward (ae_modules.py:602-641)."""
    nres = len(ddconfig["ch_mult"])
    nrb = ddconfig["num_res_blocks"]
    p = prefix
    h = _conv(sd, p + ".conv_in", z, 1)
    h = resnet_block(sd, p + ".mid.block_1", h)
    h = attn_block(sd, p + ".mid.attn_1", h)
    h = resnet_block(sd, p + ".mid.block_2", h)
    for lvl in reversed(range(nres)):
        for ib in range(nrb + 1):
            h = resnet_block(sd, f"{p}.up.{lvl}.block.{ib}", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")  # Upsample ae_modules.py:118-122
            h = _conv(sd, f"{p}.up.{lvl}.upsample.conv", h, 1)
    h = _swish(_gn(sd, p + ".norm_out", h))
    return _conv(sd, p + ".conv_out", h, 1)


@torch.no_grad()
def decode_first_stage_2dae(sd, ddconfig, z, scale_factor=0.18215):
    """z (b,4,t,h,w) -> video (b,3,t,8h,8w); frame loop of ddpm3d.py:666-679."""
    sd = {k: v.float() for k, v in sd.items()}
    z = z.float() / scale_factor
    frames = []
    for i in range(z.shape[2]):
        zi = _conv(sd, "post_quant_conv", z[:, :, i], 0)  # autoencoder.py:110-113
        frames.append(decoder_forward(sd, ddconfig, zi).unsqueeze(2))
    return torch.cat(frames, dim=2)


@torch.no_grad()
def encoder_forward(sd, ddconfig, x, prefix="encoder"):
    """Encoder.forward (ae_modules.py:470-503): conv_in, per level [ResnetBlock x n, Downsample with the
    asymmetric (0,1,0,1) zero pad (ae_modules.py:98-102)], mid, GroupNorm, swish, conv_out."""
    nres = len(ddconfig["ch_mult"])
    nrb = ddconfig["num_res_blocks"]
    p = prefix
    h = _conv(sd, p + ".conv_in", x, 1)
    for lvl in range(nres):
        for ib in range(nrb):
            h = resnet_block(sd, f"{p}.down.{lvl}.block.{ib}", h)
        if lvl != nres - 1:
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[f"{p}.down.{lvl}.downsample.conv.weight"],
                         sd[f"{p}.down.{lvl}.downsample.conv.bias"], stride=2)
    h = resnet_block(sd, p + ".mid.block_1", h)
    h = attn_block(sd, p + ".mid.attn_1", h)
    h = resnet_block(sd, p + ".mid.block_2", h)
    return _conv(sd, p + ".conv_out", _swish(_gn(sd, p + ".norm_out", h)), 1)


@torch.no_grad()
def encode_moments(sd, ddconfig, x):
    """AutoencoderKL.encode up to the posterior parameters (autoencoder.py:103-108): quant_conv(encoder(x));
    mean/logvar = chunk(2), logvar clamped to [-30, 20] (lvdm/distributions.py:24-31)."""
    sd = {k: v.float() for k, v in sd.items()}
    m = _conv(sd, "quant_conv", encoder_forward(sd, ddconfig, x.float()), 0)
    mean, logvar = torch.chunk(m, 2, dim=1)
    return m, mean, torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
