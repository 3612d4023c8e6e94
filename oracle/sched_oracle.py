"""CPU restatement of the scheduler / consistency-distillation math (oracle, test-only).

The reference's ``T2VTurboScheduler`` needs ``diffusers`` mixins that are not
installed here, so it cannot be imported; this file restates its arithmetic
(``scheduler/t2v_turbo_scheduler.py``) and the CD helpers of
``utils/common_utils.py`` / ``ode_solver/ddim_solver.py`` as plain functions.
Pinned by the closed-form known answers of SURVEY.md §8(c) (timestep tables,
boundary scalings) and by executing the reference source text of the pure
helper functions in ``tests/golden/make_golden.py``.
"""
import numpy as np
import torch


def alphas_cumprod(num_train_timesteps=1000, linear_start=0.00085, linear_end=0.012):
    """scaled_linear betas -> cumprod, fp32 (t2v_turbo_scheduler.py:213-235)."""
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def lcm_timesteps(num_inference_steps, lcm_origin_steps, num_train_timesteps=1000):
    """set_timesteps (t2v_turbo_scheduler.py:323-355)."""
    c = num_train_timesteps // lcm_origin_steps
    origin = np.asarray(list(range(1, lcm_origin_steps + 1))) * c - 1
    skip = len(origin) // num_inference_steps
    return origin[::-skip][:num_inference_steps].copy()


def boundary_scalings(t, sigma_data=0.5):
    """get_scalings_for_boundary_condition_discrete (t2v_turbo_scheduler.py:359-365)."""
    c_skip = sigma_data ** 2 / ((t / 0.1) ** 2 + sigma_data ** 2)
    c_out = (t / 0.1) / ((t / 0.1) ** 2 + sigma_data ** 2) ** 0.5
    return c_skip, c_out


def step(acp, timesteps, model_output, timeindex, timestep, sample, noise):
    """T2VTurboScheduler.step, epsilon prediction (t2v_turbo_scheduler.py:367-467).
    ``noise`` is the N(0,1) draw the reference takes from ``randn_tensor``."""
    prev_idx = timeindex + 1
    prev_t = timesteps[prev_idx] if prev_idx < len(timesteps) else timestep
    a_t = acp[int(timestep)]
    a_prev = acp[int(prev_t)] if prev_t >= 0 else torch.tensor(1.0)
    c_skip, c_out = boundary_scalings(float(timestep))
    pred_x0 = (sample - (1 - a_t).sqrt() * model_output) / a_t.sqrt()
    denoised = c_out * pred_x0 + c_skip * sample
    if len(timesteps) > 1:
        prev = a_prev.sqrt() * denoised + (1 - a_prev).sqrt() * noise
    else:
        prev = denoised
    return prev, denoised


def add_noise(acp, x0, noise, t):
    """add_noise (t2v_turbo_scheduler.py:470-495)."""
    a = acp[t].to(x0.dtype)
    sa = (a ** 0.5).reshape(-1, *([1] * (x0.dim() - 1)))
    sb = ((1 - a) ** 0.5).reshape(-1, *([1] * (x0.dim() - 1)))
    return sa * x0 + sb * noise


def w_embedding(w, embedding_dim=512, dtype=torch.float32):
    """get_w_embedding / guidance_scale_embedding: sin||cos, x1000
    (pipeline/t2v_turbo_vc2_pipeline.py:99-120; utils/common_utils.py:47-73)."""
    w = w * 1000.0
    half = embedding_dim // 2
    emb = torch.log(torch.tensor(10000.0)) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=dtype) * -emb)
    emb = w.to(dtype)[:, None] * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
    if embedding_dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1))
    return emb


def scalings_for_boundary_conditions(timestep, sigma_data=0.5, timestep_scaling=10.0):
    """utils/common_utils.py:87-91."""
    c_skip = sigma_data ** 2 / ((timestep * timestep_scaling) ** 2 + sigma_data ** 2)
    c_out = (timestep * timestep_scaling) / ((timestep * timestep_scaling) ** 2 + sigma_data ** 2) ** 0.5
    return c_skip, c_out


def _extract(a, t, ndim):
    return a.gather(-1, t).reshape(t.shape[0], *((1,) * (ndim - 1)))


def predicted_original_sample(model_output, timesteps, sample, prediction_type, alphas, sigmas):
    """get_predicted_original_sample (utils/common_utils.py:95-112)."""
    a = _extract(alphas, timesteps, sample.dim())
    s = _extract(sigmas, timesteps, sample.dim())
    if prediction_type == "epsilon":
        return (sample - s * model_output) / a
    if prediction_type == "sample":
        return model_output
    if prediction_type == "v_prediction":
        return a * sample - s * model_output
    raise ValueError(prediction_type)


def predicted_noise(model_output, timesteps, sample, prediction_type, alphas, sigmas):
    """get_predicted_noise (utils/common_utils.py:116-133)."""
    a = _extract(alphas, timesteps, sample.dim())
    s = _extract(sigmas, timesteps, sample.dim())
    if prediction_type == "epsilon":
        return model_output
    if prediction_type == "sample":
        return (sample - a * model_output) / s
    if prediction_type == "v_prediction":
        return a * model_output + s * sample
    raise ValueError(prediction_type)


def huber_loss(pred, target, huber_c=0.001):
    """utils/common_utils.py:302-304."""
    return torch.mean(torch.sqrt((pred.float() - target.float()) ** 2 + huber_c ** 2) - huber_c)


class DDIMSolverOracle:
    """DDIMSolver (ode_solver/ddim_solver.py:7-97), eta = 0."""

    def __init__(self, alpha_cumprods, timesteps=1000, ddim_timesteps=50, scale_a=1.0, scale_b=0.7,
                 mid_step=400, use_scale=False):
        self.alpha_cumprods = torch.from_numpy(alpha_cumprods)
        self.step_ratio = timesteps // ddim_timesteps
        dts = (np.arange(1, ddim_timesteps + 1) * self.step_ratio).round().astype(np.int64) - 1
        self.ddim_timesteps = torch.from_numpy(dts).long()
        self.ddim_alpha_cumprods = torch.from_numpy(alpha_cumprods[dts])
        self.ddim_alpha_cumprods_prev = torch.from_numpy(
            np.asarray([alpha_cumprods[0]] + alpha_cumprods[dts[:-1]].tolist()))
        self.use_scale = use_scale
        if use_scale:
            arr = np.concatenate((np.linspace(scale_a, scale_b, mid_step), np.full(timesteps, scale_b)))
            self.ddim_scale_arr = torch.from_numpy(arr[dts])
            self.ddim_scale_arr_prev = torch.from_numpy(np.asarray([arr[0]] + arr[dts[:-1]].tolist()))

    def ddim_step(self, pred_x0, pred_noise, timestep_index):
        a_prev = _extract(self.ddim_alpha_cumprods_prev, timestep_index, pred_x0.dim())
        dir_xt = (1.0 - a_prev).sqrt() * pred_noise
        if self.use_scale:  # eta = 0 -> the sigma_t * randn term vanishes
            coef = (_extract(self.ddim_scale_arr_prev, timestep_index, pred_x0.dim())
                    / _extract(self.ddim_scale_arr, timestep_index, pred_x0.dim()))
            return a_prev.sqrt() * coef * pred_x0 + dir_xt
        return a_prev.sqrt() * pred_x0 + dir_xt

    def ddim_reverse_step(self, x_prev, pred_noise, ts):
        prev_ts = (ts - self.step_ratio).clip(min=0)
        a_next = _extract(self.alpha_cumprods, ts, x_prev.dim())
        a = _extract(self.alpha_cumprods, prev_ts, x_prev.dim())
        return (x_prev - (1 - a).sqrt() * pred_noise) * (a_next / a).sqrt() + (1 - a_next).sqrt() * pred_noise
