"""Host-side mirror of the reference UNet: state-dict / module-tree contract and the CPU
(composite) forward vs the golden vectors produced by the reference."""
import copy

import pytest
import torch
import torch.nn as nn

from oracle.synth import synth_state_dict
from t2v_turbo_amd.unet3d import TemporalTransformer, UNetModel
from tests.util import VC2_UNET, load, manifest, rel_l2, tiny_unet_params


def _leaf_order(m):
    return [[n, type(x).__name__] for n, x in m.named_modules()
            if type(x) in (nn.Linear, nn.Conv2d, nn.Conv3d, nn.Conv1d, nn.GroupNorm, nn.LayerNorm)
            or type(x).__name__ == "GroupNormSpecific"]


def test_full_size_contract_on_meta():
    with torch.device("meta"):
        m = UNetModel(**VC2_UNET)
        mg = UNetModel(**dict(VC2_UNET, motion_cond_proj_dim=256))
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == manifest("unet_full")
    assert [[k, list(v.shape)] for k, v in mg.state_dict().items()] == manifest("unet_full_mg")
    # LoRA injection order = registration order of exact-class leaves (utils/lora.py:263-307)
    assert _leaf_order(m) == manifest("unet_full_leaf_order")
    assert type(m).__name__ == "UNetModel"
    assert sum(p.numel() for p in m.parameters()) == 1413366340
    probes = [n for n, x in m.named_modules() if n.endswith("transformer_blocks.0.attn1") and ".2." in n
              and n.startswith("output_blocks")]
    assert probes == manifest("unet_full_probe_names")
    assert any(isinstance(x, TemporalTransformer) for x in m.modules())
    assert any(n.startswith("init_attn.0") for n, _ in m.named_parameters())


def test_zero_init_is_zero_and_deepcopy():
    m = UNetModel(**tiny_unet_params()).eval()
    g = load("unet_tiny")
    with torch.no_grad():
        y = m(g["x"], g["ts"], context=g["ctx"], fps=16, timestep_cond=g["tc"])
    assert float(y.abs().max()) == 0.0  # zero_module'd output conv
    m2 = copy.deepcopy(m)
    assert m2._engine_box.engine is None and m2 is not m


def test_cpu_forward_matches_reference_golden():
    g = load("unet_tiny")
    m = UNetModel(**tiny_unet_params(record_attn_probs=True)).eval()
    m.load_state_dict(synth_state_dict(manifest("unet_tiny")), strict=True)
    with torch.no_grad():
        y = m(g["x"], g["ts"], context=g["ctx"], fps=16, timestep_cond=g["tc"])
        probs = dict(m.named_modules())["output_blocks.11.2.transformer_blocks.0.attn1"].attention_probs
        y2 = m(g["x"], g["ts"], context=g["ctx"])
    assert rel_l2(y, g["y"]) < 1e-5
    assert rel_l2(y2, g["y_nocond"]) < 1e-5
    assert rel_l2(probs, g["probs_ob11"]) < 1e-5


def test_cpu_forward_motion_cond_batch2():
    g = load("unet_tiny_mg_b2")
    m = UNetModel(**tiny_unet_params(motion_cond_proj_dim=256)).eval()
    m.load_state_dict(synth_state_dict(manifest("unet_tiny_mg_b2")), strict=True)
    with torch.no_grad():
        y = m(g["x"], g["ts"], context=g["ctx"], fps=8, timestep_cond=g["tc"], motion_cond=g["mc"])
    assert rel_l2(y, g["y"]) < 1e-5


def test_bad_args_raise():
    with pytest.raises(AssertionError):
        UNetModel(**dict(tiny_unet_params(), num_head_channels=-1))
    m = UNetModel(**tiny_unet_params(motion_cond_proj_dim=256))
    g = load("unet_tiny_mg_b2")
    with pytest.raises(AssertionError):  # motion_cond without timestep_cond (openaimodel3d.py:691)
        m(g["x"], g["ts"], context=g["ctx"], motion_cond=g["mc"])


def test_auto_route_decisions():
    """``UNetModel._auto_route`` (what a CUDA call lands on under ``native_mode = "auto"``): inference engine for no-grad eval
    calls, gradient engine for a LoRA student (grad or not, train or eval) and for the full fine-tuning student, torch composite —
    named reason — for the rest."""
    import warnings
    from t2v_turbo_amd import lora, unet3d
    from tests.util import tiny_unet_params
    m = unet3d.UNetModel(**tiny_unet_params()).eval()
    x, ctx, tc = torch.zeros(1, 4, 4, 8, 8), torch.zeros(1, 7, 128), torch.zeros(1, 256)
    with torch.no_grad():
        assert m._auto_route(x, ctx, tc, None) == ("infer", None)
    assert m._auto_route(x, ctx, tc, None) == ("train_full", None)    # every parameter trainable, no LoRA: full fine-tuning (round 6: native)
    assert m._auto_route(x, ctx.clone().requires_grad_(True), tc, None)[0] == "composite"   # ... but not with a gradient w.r.t. the context
    m.native_full = False
    assert m._auto_route(x, ctx, tc, None)[0] == "composite"          # T2V_NATIVE_FULL=0
    del m.native_full
    m.train()
    assert m._auto_route(x, ctx, tc, None) == ("train_full", None)    # the v2 student is in train mode (temporal-conv dropouts live)
    m.eval()
    m.requires_grad_(False)
    assert m._auto_route(x, ctx, tc, None) == ("infer", None)          # nothing wants a gradient
    assert m._auto_route(x.clone().requires_grad_(True), ctx, tc, None)[0] == "composite"   # input gradient without LoRA
    m.train()
    with torch.no_grad():
        # train mode, no LoRA, the only live dropouts those of the temporal conv blocks: the v1 teacher, which the reference never
        # puts in eval mode (train_t2v_turbo_v1_lora.py:621-626) — the inference engine runs it with counter-based masks (round 4)
        assert m._auto_route(x, ctx, tc, None) == ("infer", None)
    m.eval()
    lora.inject_trainable_lora_extended(m, r=4)
    m.train()
    assert m._auto_route(x, ctx, tc, None) == ("train", None)          # the student forward
    with torch.no_grad():
        assert m._auto_route(x, ctx, tc, None) == ("train", None)      # the target forward (train mode, no grad)
    assert m._auto_route(x, ctx.clone().requires_grad_(True), tc, None)[0] == "composite"
    next(p for n, p in m.named_parameters() if "lora" not in n).requires_grad_(True)
    assert m._auto_route(x, ctx, tc, None)[0] == "composite"          # a base weight trainable besides the LoRA tensors
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        unet3d._WARNED_ATEN.discard("test reason")
        unet3d._warn_aten_route("test reason")
        unet3d._warn_aten_route("test reason")
    assert len([x for x in w if "ATen" in str(x.message)]) == 1        # once per reason
