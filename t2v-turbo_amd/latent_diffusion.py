"""The slice of ``LatentDiffusion`` (reference ``lvdm/models/ddpm3d.py``) the t2v-turbo entry points
touch: ``.model.diffusion_model`` (the UNet), ``.first_stage_model`` (KL-VAE), ``.cond_stage_model``
(text encoder, optional/out of scope), ``scale_factor``, ``temporal_length`` and the
``decode_first_stage_2DAE`` / ``encode_first_stage_2DAE`` helpers.  State-dict keys keep the
checkpoint prefixes (``model.diffusion_model.*``, ``first_stage_model.*``) so VideoCrafter2's
``model.ckpt["state_dict"]`` loads with ``strict=False`` for the text tower."""
import torch
import torch.nn as nn

from .unet3d import UNetModel
from .vae import AutoencoderKL


class DiffusionWrapper(nn.Module):
    def __init__(self, diffusion_model):
        super().__init__()
        self.diffusion_model = diffusion_model


class LatentDiffusion(nn.Module):
    def __init__(self, unet_config, first_stage_config, cond_stage_model=None, scale_factor=0.18215,
                 linear_start=0.00085, linear_end=0.012, timesteps=1000, channels=4, image_size=(40, 64), **ignored):
        super().__init__()
        unet = unet_config if isinstance(unet_config, nn.Module) else UNetModel(**unet_config["params"])
        vae = first_stage_config if isinstance(first_stage_config, nn.Module) else AutoencoderKL(**first_stage_config["params"])
        self.model = DiffusionWrapper(unet)
        self.first_stage_model = vae
        self.cond_stage_model = cond_stage_model
        self.scale_factor = scale_factor
        self.channels = channels
        self.image_size = image_size
        self.temporal_length = getattr(unet, "temporal_length", 16) if hasattr(unet, "temporal_length") else 16
        self.linear_start, self.linear_end, self.num_timesteps = linear_start, linear_end, timesteps
        self.encoder_type = "2d"

    @classmethod
    def from_config(cls, config, cond_stage_model=None):
        """config: the ``model`` dict of configs/inference_t2v_512_v2.0.yaml (target/params layout)."""
        p = config["params"]
        keys = ("scale_factor", "linear_start", "linear_end", "timesteps", "channels", "image_size")
        return cls(p["unet_config"], p["first_stage_config"], cond_stage_model, **{k: p[k] for k in keys if k in p})

    @torch.no_grad()
    def decode_first_stage_2DAE(self, z, **kwargs):
        """(b,4,t,h,w) -> (b,3,t,8h,8w); all frames in one batched pass (ddpm3d.py:666-679)."""
        return self.first_stage_model.decode_video(z, self.scale_factor)

    @torch.no_grad()
    def encode_first_stage_2DAE(self, x):
        """(b,3,t,H,W) -> scaled latent sample per frame (ddpm3d.py:586-600); the posterior parameters of
        all frames come from one batched encoder pass, the reparameterised sample is drawn like the reference
        (``DiagonalGaussianDistribution.sample``: CPU randn moved to the device)."""
        from .vae import DiagonalGaussianDistribution
        moments = self.first_stage_model.encode_moments_video(x)
        return self.scale_factor * DiagonalGaussianDistribution(moments.to(x.dtype)).sample().detach()

    def get_learned_conditioning(self, c):
        if self.cond_stage_model is None:
            raise RuntimeError("no text encoder attached: pass prompt_embeds (B,77,1024) instead of prompts")
        return self.cond_stage_model(c)
