"""T2VTurboScheduler — the LCM-style multistep scheduler of the reference
(``scheduler/t2v_turbo_scheduler.py``) without the diffusers mixins: same constructor arguments,
``config`` attribute access, ``alphas_cumprod`` / ``timesteps`` / ``init_noise_sigma`` members and
``set_timesteps`` / ``step`` / ``add_noise`` / ``get_velocity`` semantics.

``step`` on CUDA fp32 latents runs as ONE fused HIP kernel (x0 prediction, boundary-condition blend,
re-noising: ``t2v_lcm_step``); elsewhere it is the same arithmetic in torch."""
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch


class _Config(dict):
    __getattr__ = dict.__getitem__


@dataclass
class T2VTurboSchedulerOutput:
    prev_sample: torch.Tensor
    denoised: Optional[torch.Tensor] = None


class T2VTurboScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, linear_start=0.00085, linear_end=0.012, beta_schedule="scaled_linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                 clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False):
        assert beta_schedule == "scaled_linear"
        assert trained_betas is None
        self.config = _Config(
            num_train_timesteps=num_train_timesteps, linear_start=linear_start, linear_end=linear_end,
            beta_schedule=beta_schedule, trained_betas=trained_betas, clip_sample=clip_sample,
            set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, prediction_type=prediction_type,
            thresholding=thresholding, dynamic_thresholding_ratio=dynamic_thresholding_ratio,
            clip_sample_range=clip_sample_range, sample_max_value=sample_max_value,
            timestep_spacing=timestep_spacing, rescale_betas_zero_snr=rescale_betas_zero_snr)
        self.betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self.sigma_data = 0.5
        self._ops = None

    def __len__(self):
        return self.config.num_train_timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, lcm_origin_steps, device=None):
        if num_inference_steps > self.config.num_train_timesteps:
            raise ValueError(
                f"`num_inference_steps`: {num_inference_steps} cannot be larger than `self.config.train_timesteps`:"
                f" {self.config.num_train_timesteps} as the unet model trained with this scheduler can only handle"
                f" maximal {self.config.num_train_timesteps} timesteps.")
        self.num_inference_steps = num_inference_steps
        c = self.config.num_train_timesteps // lcm_origin_steps
        origin = np.arange(1, lcm_origin_steps + 1) * c - 1           # LCM training grid
        skip = len(origin) // num_inference_steps
        self.timesteps = torch.from_numpy(origin[::-skip][:num_inference_steps].copy()).to(device)

    def get_scalings_for_boundary_condition_discrete(self, t):
        self.sigma_data = 0.5
        c_skip = self.sigma_data ** 2 / ((t / 0.1) ** 2 + self.sigma_data ** 2)
        c_out = (t / 0.1) / ((t / 0.1) ** 2 + self.sigma_data ** 2) ** 0.5
        return c_skip, c_out

    def step(self, model_output, timeindex, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        prev_idx = timeindex + 1
        prev_timestep = self.timesteps[prev_idx] if prev_idx < len(self.timesteps) else timestep
        a_t = self.alphas_cumprod[int(timestep)]
        a_prev = self.alphas_cumprod[int(prev_timestep)] if prev_timestep >= 0 else self.final_alpha_cumprod
        c_skip, c_out = self.get_scalings_for_boundary_condition_discrete(float(timestep))
        multi = len(self.timesteps) > 1
        noise = None
        if multi:
            if variance_noise is not None:
                noise = variance_noise
            else:  # diffusers randn_tensor: draw on the generator's device, then move
                gdev = generator.device if generator is not None else model_output.device
                noise = torch.randn(model_output.shape, generator=generator, device=gdev,
                                    dtype=model_output.dtype).to(model_output.device)
        kind = self.config.prediction_type
        if (kind == "epsilon" and sample.is_cuda and sample.dtype == torch.float32 and sample.is_contiguous()
                and model_output.is_contiguous() and not torch.is_grad_enabled()):
            prev, den = torch.empty_like(sample), torch.empty_like(sample)
            self._hip().lcm_step(sample, model_output, noise.float().contiguous() if multi else None,
                                 float(a_t.sqrt()), float((1 - a_t).sqrt()), float(c_skip), float(c_out),
                                 float(a_prev.sqrt()), float((1 - a_prev).sqrt()), prev, den)
        else:
            if kind == "epsilon":
                x0 = (sample - (1 - a_t).sqrt() * model_output) / a_t.sqrt()
            elif kind == "sample":
                x0 = model_output
            elif kind == "v_prediction":
                x0 = a_t.sqrt() * sample - (1 - a_t).sqrt() * model_output
            else:
                raise ValueError(kind)
            den = c_out * x0 + c_skip * sample
            prev = a_prev.sqrt() * den + (1 - a_prev).sqrt() * noise if multi else den
        if not return_dict:
            return (prev, den)
        return T2VTurboSchedulerOutput(prev_sample=prev, denoised=den)

    def _hip(self):
        if self._ops is None:
            from .native import HipOps
            self._ops = HipOps()
        return self._ops

    def _acp_on(self, device, dtype):
        """alphas_cumprod on the samples' device, copied once (a pageable host -> device copy per call synchronises the stream)."""
        key = (str(device), dtype)
        hit = getattr(self, "_acp_cache", None)
        if hit is None or hit[0] != key or hit[1] is not self.alphas_cumprod:
            hit = self._acp_cache = (key, self.alphas_cumprod, self.alphas_cumprod.to(device=device, dtype=dtype))
        return hit[2]

    def add_noise(self, original_samples, noise, timesteps):
        acp = self._acp_on(original_samples.device, original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        shape = (-1,) + (1,) * (original_samples.dim() - 1)
        return (acp[timesteps] ** 0.5).reshape(shape) * original_samples + ((1 - acp[timesteps]) ** 0.5).reshape(shape) * noise

    def get_velocity(self, sample, noise, timesteps):
        acp = self._acp_on(sample.device, sample.dtype)
        timesteps = timesteps.to(sample.device)
        shape = (-1,) + (1,) * (sample.dim() - 1)
        return (acp[timesteps] ** 0.5).reshape(shape) * noise - ((1 - acp[timesteps]) ** 0.5).reshape(shape) * sample
