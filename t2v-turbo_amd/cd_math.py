"""Consistency-distillation tensor math of the trainers (reference ``utils/common_utils.py:47-133,
302-319`` and ``ode_solver/ddim_solver.py``): boundary scalings, x0/eps conversions, guidance
embedding, pseudo-Huber loss, EMA update and the teacher DDIM solver.  Plain torch: these sit on the
autograd path of training (the no-grad sampling step has its fused HIP kernel in scheduler.py)."""
import numpy as np
import torch

from .nn_util import guidance_embedding as guidance_scale_embedding  # noqa: F401  (same function)


def extract_into_tensor(a, t, x_shape):
    b, *_ = t.shape
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))


def append_dims(x, target_dims):
    extra = target_dims - x.ndim
    if extra < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * extra]


def scalings_for_boundary_conditions(timestep, sigma_data=0.5, timestep_scaling=10.0):
    s = timestep_scaling * timestep
    return sigma_data ** 2 / (s ** 2 + sigma_data ** 2), s / (s ** 2 + sigma_data ** 2) ** 0.5


def get_predicted_original_sample(model_output, timesteps, sample, prediction_type, alphas, sigmas):
    a = extract_into_tensor(alphas, timesteps, sample.shape)
    s = extract_into_tensor(sigmas, timesteps, sample.shape)
    if prediction_type == "epsilon":
        return (sample - s * model_output) / a
    if prediction_type == "sample":
        return model_output
    if prediction_type == "v_prediction":
        return a * sample - s * model_output
    raise ValueError(f"Prediction type {prediction_type} is not supported; currently, `epsilon`, `sample`, and "
                     f"`v_prediction` are supported.")


def get_predicted_noise(model_output, timesteps, sample, prediction_type, alphas, sigmas):
    a = extract_into_tensor(alphas, timesteps, sample.shape)
    s = extract_into_tensor(sigmas, timesteps, sample.shape)
    if prediction_type == "epsilon":
        return model_output
    if prediction_type == "sample":
        return (sample - a * model_output) / s
    if prediction_type == "v_prediction":
        return a * model_output + s * sample
    raise ValueError(f"Prediction type {prediction_type} is not supported; currently, `epsilon`, `sample`, and "
                     f"`v_prediction` are supported.")


def huber_loss(pred, target, huber_c=0.001):
    return (torch.sqrt((pred.float() - target.float()) ** 2 + huber_c ** 2) - huber_c).mean()


@torch.no_grad()
def update_ema(target_params, source_params, rate=0.99):
    for targ, src in zip(target_params, source_params):
        targ.detach().mul_(rate).add_(src.to(targ.dtype), alpha=1 - rate)


class DDIMSolver:
    """Teacher ODE step on the 50/200-point DDIM grid (ode_solver/ddim_solver.py:7-97)."""

    def __init__(self, alpha_cumprods, timesteps=1000, ddim_timesteps=50, scale_a=1.0, scale_b=0.7, mid_step=400,
                 ddim_eta=0.0, use_scale=False):
        self.alpha_cumprods = torch.from_numpy(alpha_cumprods)
        self.step_ratio = timesteps // ddim_timesteps
        grid = (np.arange(1, ddim_timesteps + 1) * self.step_ratio).round().astype(np.int64) - 1
        self.ddim_timesteps = torch.from_numpy(grid).long()
        self.ddim_alpha_cumprods = torch.from_numpy(alpha_cumprods[grid])
        self.ddim_alpha_cumprods_prev = torch.from_numpy(np.asarray([alpha_cumprods[0]] + alpha_cumprods[grid[:-1]].tolist()))
        self.use_scale = use_scale
        if use_scale:  # VideoCrafter2's scale schedule (incl. its constant second half)
            arr = np.concatenate((np.linspace(scale_a, scale_b, mid_step), np.full(timesteps, scale_b)))
            self.ddim_scale_arr = torch.from_numpy(arr[grid])
            self.ddim_scale_arr_prev = torch.from_numpy(np.asarray([arr[0]] + arr[grid[:-1]].tolist()))
            self.ddim_sigmas = ddim_eta * torch.sqrt((1 - self.ddim_alpha_cumprods_prev) / (1 - self.ddim_alpha_cumprods)
                                                     * (1 - self.ddim_alpha_cumprods / self.ddim_alpha_cumprods_prev))

    def to(self, device, dtype=None):
        self.alpha_cumprods = self.alpha_cumprods.to(device, dtype)
        self.ddim_timesteps = self.ddim_timesteps.to(device)
        self.ddim_alpha_cumprods = self.ddim_alpha_cumprods.to(device, dtype)
        self.ddim_alpha_cumprods_prev = self.ddim_alpha_cumprods_prev.to(device, dtype)
        if self.use_scale:
            self.ddim_scale_arr = self.ddim_scale_arr.to(device, dtype)
            self.ddim_scale_arr_prev = self.ddim_scale_arr_prev.to(device, dtype)
            self.ddim_sigmas = self.ddim_sigmas.to(device, dtype)
        return self

    def ddim_step(self, pred_x0, pred_noise, timestep_index):
        a_prev = extract_into_tensor(self.ddim_alpha_cumprods_prev, timestep_index, pred_x0.shape)
        dir_xt = (1.0 - a_prev).sqrt() * pred_noise
        if self.use_scale:
            coef = (extract_into_tensor(self.ddim_scale_arr_prev, timestep_index, pred_x0.shape)
                    / extract_into_tensor(self.ddim_scale_arr, timestep_index, pred_x0.shape))
            noise = extract_into_tensor(self.ddim_sigmas, timestep_index, pred_x0.shape) * torch.randn_like(pred_x0)
            return a_prev.sqrt() * coef * pred_x0 + dir_xt + noise
        return a_prev.sqrt() * pred_x0 + dir_xt

    def ddim_reverse_step(self, x_prev, pred_noise, ts):
        assert not self.use_scale
        prev_ts = (ts - self.step_ratio).clip(min=0)
        a_next = extract_into_tensor(self.alpha_cumprods, ts, x_prev.shape)
        a = extract_into_tensor(self.alpha_cumprods, prev_ts, x_prev.shape)
        return (x_prev - (1 - a).sqrt() * pred_noise) * (a_next / a).sqrt() + (1 - a_next).sqrt() * pred_noise
