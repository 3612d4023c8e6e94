"""T2VTurboVC2Pipeline — the few-step sampling loop of the reference
(``pipeline/t2v_turbo_vc2_pipeline.py:122-220``) around the native UNet, the fused scheduler step and
the batched VAE decode.  Same constructor and ``__call__`` keyword surface; no diffusers dependency
(``register_modules`` / ``progress_bar`` / ``_execution_device`` plumbing is local)."""
from typing import Any, Dict, List, Optional, Union

import torch

from .nn_util import guidance_embedding
from .scheduler import T2VTurboScheduler


class T2VTurboVC2Pipeline:
    def __init__(self, pretrained_t2v, scheduler: Optional[T2VTurboScheduler] = None, model_config: Dict[str, Any] = None):
        self.pretrained_t2v = pretrained_t2v
        self.scheduler = scheduler if scheduler is not None else T2VTurboScheduler(
            linear_start=getattr(pretrained_t2v, "linear_start", 0.00085),
            linear_end=getattr(pretrained_t2v, "linear_end", 0.012))
        self.vae = pretrained_t2v.first_stage_model
        self.unet = pretrained_t2v.model.diffusion_model
        self.text_encoder = pretrained_t2v.cond_stage_model
        self.model_config = model_config
        self.vae_scale_factor = 8

    # -- diffusers-style plumbing ---------------------------------------------------------------------
    def to(self, *args, **kwargs):
        self.pretrained_t2v.to(*args, **kwargs)
        return self

    @property
    def device(self):
        return next(self.unet.parameters()).device

    _execution_device = device

    @property
    def dtype(self):
        return next(self.unet.parameters()).dtype

    def _encode_prompt(self, prompt, device, num_videos_per_prompt, prompt_embeds=None):
        if prompt_embeds is None:
            if self.text_encoder is None:
                raise RuntimeError("no text encoder attached: pass prompt_embeds (B,77,1024)")
            prompt_embeds = self.text_encoder(prompt)
        prompt_embeds = prompt_embeds.to(device=device)
        bs, seq, _ = prompt_embeds.shape
        return prompt_embeds.repeat(1, num_videos_per_prompt, 1).view(bs * num_videos_per_prompt, seq, -1)

    def prepare_latents(self, batch_size, num_channels_latents, frames, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, num_channels_latents, frames, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            gdev = generator.device if generator is not None else device
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        else:
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    get_w_embedding = staticmethod(guidance_embedding)

    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str]] = None, height: Optional[int] = 320, width: Optional[int] = 512,
                 frames: int = 16, fps: int = 16, guidance_scale: float = 7.5, motion_gs: float = 0.1,
                 use_motion_cond: bool = False, percentage: float = 0.3, num_videos_per_prompt: Optional[int] = 1,
                 generator=None, latents: Optional[torch.Tensor] = None, num_inference_steps: int = 4,
                 lcm_origin_steps: int = 50, prompt_embeds: Optional[torch.Tensor] = None,
                 output_type: Optional[str] = "pil"):
        unet_params = self.model_config["params"]["unet_config"]["params"]
        frames = self.pretrained_t2v.temporal_length if frames < 0 else frames
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        device = self.device
        prompt_embeds = self._encode_prompt(prompt, device, num_videos_per_prompt, prompt_embeds=prompt_embeds)
        self.scheduler.set_timesteps(num_inference_steps, lcm_origin_steps)
        timesteps = self.scheduler.timesteps
        bs = batch_size * num_videos_per_prompt
        latents = self.prepare_latents(bs, unet_params["in_channels"], frames, height, width, prompt_embeds.dtype,
                                       device, generator, latents)
        context = {"context": prompt_embeds.to(self.dtype), "fps": fps}
        w = torch.tensor(guidance_scale).repeat(bs)
        context["timestep_cond"] = guidance_embedding(w, embedding_dim=256).to(device).to(self.dtype)
        ms_t_threshold = self.scheduler.config.num_train_timesteps * (1 - percentage)
        denoised = latents
        if use_motion_cond:
            # the motion-guidance embedding takes two values over the whole trajectory (motion_gs above the threshold, 0 below:
            # pipeline/t2v_turbo_vc2_pipeline.py:190-204): both are made ONCE and put on the device here — a host-to-device copy inside
            # the loop is a host / stream synchronisation per step (1 ms of launch overhead per step no longer hidden under the GPU's work)
            mg_on, mg_off = (guidance_embedding(torch.full((bs,), float(v)), embedding_dim=256, dtype=torch.float32).to(device).to(self.dtype)
                             for v in (motion_gs, 0.0))
        for i, t in enumerate(timesteps):
            ts = torch.full((bs,), int(t), device=device, dtype=torch.long)
            if use_motion_cond:
                context["motion_cond"] = mg_off if t < ms_t_threshold else mg_on
            model_pred = self.unet(latents, ts, **context)
            latents, denoised = self.scheduler.step(model_pred, i, t, latents, generator=generator, return_dict=False)
        if output_type == "latent":
            return denoised
        return self.pretrained_t2v.decode_first_stage_2DAE(denoised)


def make_synthetic_t2v(unet, device, dtype, ddconfig=None):
    """A LatentDiffusion around an existing UNet with a random-init full-size KL-VAE (benchmarks)."""
    from .latent_diffusion import LatentDiffusion
    from .vae import AutoencoderKL
    dd = ddconfig or dict(double_z=True, z_channels=4, resolution=512, in_channels=3, out_ch=3, ch=128,
                          ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    with torch.device(device):
        vae = AutoencoderKL(ddconfig=dd, embed_dim=4)
    vae = vae.to(dtype).eval()
    return LatentDiffusion(unet, vae)
