"""Motion-prior preprocessing around the denoiser: DDIM inversion — a pure consumer of the
native UNet forward, ``len(solver.ddim_timesteps)`` sequential calls — temporal attention-probability capture and the
motion-prior score (``motion_prior_sample.py:27-84``, ``utils/common_utils.py:446-478``)."""
import torch
import torch.nn.functional as F

ATTN_PROB_BLOCKS = tuple(f"output_blocks.{i}.2" for i in range(3, 12))


@torch.no_grad()
def reverse_ddim_loop(latents, unet, context, solver, num_inference_steps, device):
    """motion_prior_sample.py:27-37: x_{t_i} from x_{t_{i-1}} with the model's own noise prediction (no grad: every call
    runs on the native engine when the UNet is on a GPU)."""
    intermediate_latents = []
    for i in range(num_inference_steps):
        index = torch.full((1,), i, device=device, dtype=torch.long)
        ts = solver.ddim_timesteps[index].long()
        pred_noise = unet(latents, ts, **context)
        latents = solver.ddim_reverse_step(latents, pred_noise, ts)
        intermediate_latents.append(latents)
    return intermediate_latents


def get_temp_attn_prob(unet, latent, ts, context):
    """motion_prior_sample.py:40-57: forward + the temporal attn1 probabilities of output_blocks 3..11 (the UNet must have
    been built with ``record_attn_probs=True``)."""
    model_output = unet(latent, ts, **context)
    attention_prob = {}
    for name, module in unet.named_modules():
        if name.startswith(ATTN_PROB_BLOCKS) and name.endswith("blocks.0.attn1"):
            attention_prob[name] = module.attention_probs
    return model_output, attention_prob


def calculate_motion_rank_new(tensor_ref, tensor_gen, rank_k=1):
    """utils/common_utils.py:446-461: MSE on the top-``rank_k`` entries (by reference value) of each probability row."""
    if rank_k == 0:
        return torch.tensor(0.0, device=tensor_ref.device)
    if rank_k > tensor_ref.shape[-1]:
        raise ValueError("The value of rank_k cannot be larger than the number of frames")
    _, sorted_indices = torch.sort(tensor_ref, dim=-1)
    mask = torch.zeros_like(tensor_ref, dtype=torch.bool)
    mask.scatter_(-1, sorted_indices[..., -rank_k:], True)
    return F.mse_loss(tensor_ref[mask].detach(), tensor_gen[mask])


def compute_temp_loss(attention_prob, attention_prob_example):
    """utils/common_utils.py:464-478."""
    losses = [calculate_motion_rank_new(attention_prob_example[name].detach(), attention_prob[name], rank_k=1)
              for name in attention_prob.keys()]
    return (torch.stack(losses) * 100).mean()


def get_motion_prior_score(unet, latents, ts, example_latent, original_context, inference_context, temp_loss_scale):
    """motion_prior_sample.py:59-84: d(temp loss)/d(latents).  The example pass needs no gradient (native engine); the
    differentiated pass runs the module's torch path (the native path is forward-only and steps aside by itself when the
    input requires grad)."""
    with torch.no_grad():
        _, attention_prob_example = get_temp_attn_prob(unet, example_latent, ts, original_context)
    with torch.set_grad_enabled(True):
        latents.requires_grad_(True)
        cond_teacher_output, attention_prob = get_temp_attn_prob(unet, latents, ts, inference_context)
        loss = temp_loss_scale * compute_temp_loss(attention_prob, attention_prob_example)
        score = torch.autograd.grad(loss, latents)[0].detach()
    return score, cond_teacher_output


def get_motion_prior_score_native(grad_engine, unet, latents, ts, example_latent, original_context, inference_context,
                                  temp_loss_scale):
    """Same result as ``get_motion_prior_score`` with the differentiated pass on the UNet data-gradient engine
    (``engine_unet_bwd.UNetGradEngine``): forward with tape, the (tiny) loss on the recorded probabilities differentiated by
    autograd w.r.t. those probabilities only, and the engine's backward from there to the latents."""
    with torch.no_grad():
        _, attention_prob_example = get_temp_attn_prob(unet, example_latent, ts, original_context)
        attention_prob_example = {k: v.clone() for k, v in attention_prob_example.items()}  # the next forward reuses the buffers
    ctx = dict(inference_context)
    out = grad_engine.forward_tape(latents.detach(), ts, ctx["context"], ctx.get("fps", 16), ctx.get("timestep_cond"),
                                   ctx.get("motion_cond"))
    modules = dict(unet.named_modules())
    names = [n for n in attention_prob_example if n in modules]
    leaves = {n: modules[n].attention_probs.detach().clone().requires_grad_(True) for n in names}
    with torch.set_grad_enabled(True):
        loss = temp_loss_scale * compute_temp_loss(leaves, attention_prob_example)
        grads = torch.autograd.grad(loss, [leaves[n] for n in names])
    score = grad_engine.backward(None, {modules[n]: g for n, g in zip(names, grads)})
    return score.detach(), out

