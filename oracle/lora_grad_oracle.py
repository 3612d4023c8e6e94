"""fp32 CPU reference for the LoRA student's forward + backward (oracle, TEST INFRASTRUCTURE ONLY).

What the reference computes at ``train_t2v_turbo_v1_lora.py:1022-1028`` (student forward through the
``LoraInjected*`` leaves, ``utils/lora.py:45-50,124-129,204-209``) and ``:1190`` (``accelerator.backward``):
torch autograd on the CPU, fp32, through the reference-shaped module mirror ``t2v_turbo_amd.unet3d.UNetModel``
with ``native_mode = "off"`` (its plain torch path; no native kernel is involved).  That path is pinned to
gradients THE REFERENCE ITSELF produced — ``tests/golden/unet_tiny_lora_grad.npz`` (made by
``tests/golden/make_golden_lora_grad.py`` from the reference ``UNetModel`` + the reference
``inject_trainable_lora_extended``), checked by ``tests/test_unet_lora_grad_cpu.py::test_lora_gradients_match_the_reference_fixture``.
Activation checkpointing (``lvdm/common.py:96-112``) bounds the host memory at the full VideoCrafter2 widths.

Used by ``tests/test_gpu_train_parity.py`` (16 frames) and by ``bench.py``'s distillation-leg parity gate (a bounded
4-frame sample) as the checker of the device gradient engine; never by the product path."""
import torch


def student_reference(state_dict, unet_cfg, rank, x, ts, ctx, fps, tc, r_out, threads=None, checkpoint=True, prepare=None):
    """``state_dict``: of the LoRA-injected student (any device / dtype).  Returns (y, d<y, r_out>/dx, [d/d(lora tensor)]) on
    the CPU in fp32, LoRA tensors in ``lora.lora_parameters`` order (up0, down0, up1, down1, ...), eval mode.
    ``prepare(ref)``: called on the fp32 CPU module before the forward — the train-mode gate puts the module in train mode there
    and replaces every ``nn.Dropout`` the device engine applied by that engine's own mask (tests/mask_replay.py); with dropout
    modules replaced by fixed masks the checkpointed recomputation is the same function, so checkpointing stays on."""
    from t2v_turbo_amd import lora
    from t2v_turbo_amd.unet3d import UNetModel
    if threads:
        torch.set_num_threads(threads)
    sd = {k: v.detach().to("cpu", torch.float32) for k, v in state_dict.items()}
    with torch.device("meta"):   # no parameter initialisation: every tensor comes from the state dict
        ref = UNetModel(**dict(unet_cfg, use_checkpoint=bool(checkpoint)))
        ref.requires_grad_(False)
        lora.inject_trainable_lora_extended(ref, r=rank)
    ref = ref.to_empty(device="cpu")
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    ref.native_mode = "off"
    params = lora.lora_parameters(ref)
    for p in params:
        p.requires_grad_(True)
    if prepare is not None:
        prepare(ref)
    xg = x.detach().to("cpu", torch.float32).clone().requires_grad_(True)
    y = ref(xg, ts.cpu(), context=ctx.detach().to("cpu", torch.float32), fps=fps,
            timestep_cond=None if tc is None else tc.detach().to("cpu", torch.float32))
    (y * r_out.detach().to("cpu", torch.float32)).sum().backward()
    return y.detach(), xg.grad, [p.grad for p in params]


def per_tensor_agreement(grads, refs):
    """[(index, cosine, norm ratio)] for every LoRA tensor whose reference gradient is not identically zero (those are
    returned separately as indices: the device gradient must be ~0 there)."""
    rows, zeros = [], []
    for i, (g, r) in enumerate(zip(grads, refs)):
        g, r = g.detach().double().flatten().cpu(), r.detach().double().flatten().cpu()
        rn = float(r.norm())
        if rn == 0.0:
            zeros.append(i)
            continue
        gn = float(g.norm())
        rows.append((i, float(g @ r) / max(gn * rn, 1e-300), gn / rn))
    return rows, zeros
