"""KL-VAE (first stage) behind the reference's interface.

Mirrors ``lvdm/models/autoencoder.py`` (``AutoencoderKL.encode/decode``) and the
``Encoder`` / ``Decoder`` conv stacks of ``lvdm/modules/networks/ae_modules.py`` with identical
child names and state-dict keys.  ``decode`` of CUDA tensors without autograd runs on the gfx950
HIP engine with all frames batched in one pass (the reference decodes frame by frame,
``lvdm/models/ddpm3d.py:666-679``); CPU tensors / autograd use the torch composite below.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .nn_util import EngineBox


def Normalize(in_channels, num_groups=32):
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


def swish(x):
    return x * torch.sigmoid(x)


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.use_conv_shortcut = conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        if temb_channels > 0:
            self.temb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            if conv_shortcut:
                self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
            else:
                self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)

    def forward(self, x, temb=None):
        h = self.conv1(swish(self.norm1(x)))
        if temb is not None:
            h = h + self.temb_proj(swish(temb))[:, :, None, None]
        h = self.conv2(self.dropout(swish(self.norm2(h))))
        if self.in_channels != self.out_channels:
            x = self.conv_shortcut(x) if self.use_conv_shortcut else self.nin_shortcut(x)
        return x + h


class AttnBlock(nn.Module):
    """Single-head attention over the h*w positions, head dim = channels (ae_modules.py:29-73)."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, 1)
        self.k = nn.Conv2d(in_channels, in_channels, 1)
        self.v = nn.Conv2d(in_channels, in_channels, 1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, 1)

    def forward(self, x):
        h_ = self.norm(x)
        q, k, v = self.q(h_), self.k(h_), self.v(h_)
        b, c, h, w = q.shape
        q = q.flatten(2).transpose(1, 2)
        w_ = torch.bmm(q, k.flatten(2)) * (int(c) ** -0.5)
        w_ = F.softmax(w_, dim=2)
        h_ = torch.bmm(v.flatten(2), w_.transpose(1, 2)).reshape(b, c, h, w)
        return x + self.proj_out(h_)


def make_attn(in_channels, attn_type="vanilla"):
    assert attn_type in ("vanilla", "none"), f"attn_type {attn_type} not supported"
    return AttnBlock(in_channels) if attn_type == "vanilla" else nn.Identity()


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        self.in_channels = in_channels
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, 3, 1, 1)

    def forward(self, x):
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        return self.conv(x) if self.with_conv else x


class Downsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        self.in_channels = in_channels
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, 3, 2, 0)

    def forward(self, x):
        if self.with_conv:  # asymmetric zero pad (right/bottom), ae_modules.py:98-102
            return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))
        return F.avg_pool2d(x, kernel_size=2, stride=2)


class Encoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.conv_in = nn.Conv2d(in_channels, ch, 3, 1, 1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.in_ch_mult = in_ch_mult
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res //= 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, 3, 1, 1)

    def forward(self, x):
        h = self.conv_in(x)
        for i_level in range(self.num_resolutions):
            for i_block in range(self.num_res_blocks):
                h = self.down[i_level].block[i_block](h, None)
                if len(self.down[i_level].attn) > 0:
                    h = self.down[i_level].attn[i_block](h)
            if i_level != self.num_resolutions - 1:
                h = self.down[i_level].downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h, None)), None)
        return self.conv_out(swish(self.norm_out(h)))


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.give_pre_end, self.tanh_out = give_pre_end, tanh_out
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, 1, 1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        ups = []
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res *= 2
            ups.insert(0, up)
        self.up = nn.ModuleList(ups)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, 1, 1)

    def forward(self, z):
        self.last_z_shape = z.shape
        h = self.conv_in(z)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h, None)), None)
        for i_level in reversed(range(self.num_resolutions)):
            for i_block in range(self.num_res_blocks + 1):
                h = self.up[i_level].block[i_block](h, None)
                if len(self.up[i_level].attn) > 0:
                    h = self.up[i_level].attn[i_block](h)
            if i_level != 0:
                h = self.up[i_level].upsample(h)
        if self.give_pre_end:
            return h
        h = self.conv_out(swish(self.norm_out(h)))
        return torch.tanh(h) if self.tanh_out else h


class DiagonalGaussianDistribution:
    """Posterior of the KL-VAE (lvdm/distributions.py:24-42)."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, noise=None):
        if noise is None:
            noise = torch.randn(self.mean.shape)
        return self.mean + self.std * noise.to(device=self.parameters.device)

    def mode(self):
        return self.mean


class _NativeDecodeGrad(torch.autograd.Function):
    """decode(z) with a native backward: forward and dX both run on the HIP engine (engine_vae_bwd.py)."""

    @staticmethod
    def forward(ctx, z, vae):
        ctx.vae = vae
        ctx.z_dtype = z.dtype
        eng = vae.native_grad_engine()
        out = eng.decode_frames_tape(z.detach().unsqueeze(2), scale=1.0).squeeze(2)
        ctx.plan = eng._last
        ctx.fwd_id = eng._last["fwd_id"]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        eng = ctx.vae.native_grad_engine()
        if ctx.plan["fwd_id"] != ctx.fwd_id:
            raise RuntimeError("native VAE decode gradient keeps ONE outstanding decode per input shape: another decode of "
                               "the same shape ran before this backward (set vae.native_mode = 'off' for the torch path)")
        eng._last = ctx.plan
        dz = eng.backward(grad_out.unsqueeze(2)).squeeze(2)
        return dz.to(ctx.z_dtype), None


class AutoencoderKL(nn.Module):
    """encode(x) -> posterior; decode(z) -> image (lvdm/models/autoencoder.py:13-113)."""

    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=(), image_key="image",
                 colorize_nlabels=None, monitor=None, test=False, logdir=None, input_dim=4, test_args=None):
        super().__init__()
        assert ddconfig["double_z"]
        self.image_key = image_key
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.loss = nn.Identity()
        self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim
        self.input_dim = input_dim
        self._engine_box = EngineBox()

    @property
    def dtype(self):
        return self.post_quant_conv.weight.dtype

    def _native_ok(self, t):
        if getattr(self, "native_mode", "auto") == "off":  # "off": every entry point takes the torch path
            return False
        return t.is_cuda and not (torch.is_grad_enabled() and (t.requires_grad or any(p.requires_grad for p in self.parameters())))

    def encode(self, x, **kwargs):
        if self._native_ok(x):
            return DiagonalGaussianDistribution(self.encode_moments_video(x.unsqueeze(2)).squeeze(2).to(x.dtype))
        return DiagonalGaussianDistribution(self.quant_conv(self.encoder(x)))

    def encode_moments_video(self, x):
        """(b,3,t,H,W) -> posterior parameters (b, 2*embed, t, H/8, W/8) for all frames in one batched pass
        (the reference encodes 8-frame chunks, train_t2v_turbo_v1_lora.py:959-966)."""
        if self._native_ok(x):
            if self._engine_box.enc is None:
                from .engine_vae import VAEEncodeEngine
                from .native import HipOps
                self._engine_box.enc = VAEEncodeEngine(self, HipOps())
            return self._engine_box.enc.encode_frames(x)
        return torch.stack([self.quant_conv(self.encoder(x[:, :, i])) for i in range(x.shape[2])], dim=2)

    def decode(self, z, **kwargs):
        if getattr(self, "native_mode", "auto") == "off":
            return self.decoder(self.post_quant_conv(z))
        if z.is_cuda and not (torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for p in self.parameters()))):
            return self.native_engine().decode_frames(z.unsqueeze(2), scale=1.0).squeeze(2)
        if (z.is_cuda and torch.is_grad_enabled() and z.requires_grad and not any(p.requires_grad for p in self.parameters())
                and not self.training and getattr(self, "native_mode", "auto") != "off"):
            # reward-gradient branch (train_t2v_turbo_v1_lora.py:1047-1098): frozen VAE, gradient w.r.t. the latents only
            return _NativeDecodeGrad.apply(z, self)
        return self.decoder(self.post_quant_conv(z))

    def native_grad_engine(self):
        if self._engine_box.grad is None:
            from .engine_vae_bwd import VAEDecodeGradEngine
            from .native import HipOps
            self._engine_box.grad = VAEDecodeGradEngine(self, HipOps())
        return self._engine_box.grad

    def native_engine(self):
        if self._engine_box.engine is None:
            from .engine_vae import VAEDecodeEngine
            from .native import HipOps
            self._engine_box.engine = VAEDecodeEngine(self, HipOps())
        return self._engine_box.engine

    def decode_video(self, z, scale_factor=0.18215):
        """(b,4,t,h,w) latents -> (b,3,t,8h,8w): ``LatentDiffusion.decode_first_stage_2DAE``
        (lvdm/models/ddpm3d.py:666-679) with every frame in one batched pass on the GPU."""
        if (z.is_cuda and getattr(self, "native_mode", "auto") != "off"
                and not (torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for p in self.parameters())))):
            return self.native_engine().decode_frames(z, scale=1.0 / scale_factor)
        z = z / scale_factor
        return torch.cat([self.decode(z[:, :, i]).unsqueeze(2) for i in range(z.shape[2])], dim=2)

    def forward(self, input, sample_posterior=True):
        posterior = self.encode(input)
        z = posterior.sample() if sample_posterior else posterior.mode()
        return self.decode(z), posterior
