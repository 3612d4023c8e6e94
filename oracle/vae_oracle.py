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
