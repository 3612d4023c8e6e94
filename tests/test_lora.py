"""LoRA contract: injection census on the full-size UNet, forward equivalences,
collapse / remove, and the native engine's on-the-fly merge (via the CPU emulation backend)."""
import torch
import torch.nn as nn

from oracle.synth import synth_state_dict
from t2v_turbo_amd import lora
from t2v_turbo_amd.engine import UNetEngine
from t2v_turbo_amd.unet3d import UNetModel
from tests.emu_ops import EmuOps
from tests.util import VC2_UNET, load, manifest, rel_l2, tiny_unet_params


def test_full_size_injection_census():
    with torch.device("meta"):
        m = UNetModel(**VC2_UNET)
        params, names = lora.inject_trainable_lora_extended(m, r=64)
    kinds = [type(x).__name__ for x in m.modules() if isinstance(x, lora._INJECTED)]
    assert kinds.count("LoraInjectedLinear") == 421
    assert kinds.count("LoraInjectedConv2d") == 66
    assert kinds.count("LoraInjectedConv3d") == 88
    tensors = lora.lora_parameters(m)
    assert len(tensors) == 1150
    assert sum(t.numel() for t in tensors) == 117142176  # 468.6 MB of fp32 gradients per step
    assert len(params) == 1150 and len(names) == 575
    # rank clamps to 4 on the 4-channel in/out convs
    assert m.input_blocks[0][0].r == 4 and m.out[2].r == 4


def _tiny(with_lora=True):
    g = load("unet_tiny")
    m = UNetModel(**tiny_unet_params()).eval()
    m.load_state_dict(synth_state_dict(manifest("unet_tiny")), strict=True)
    if with_lora:
        lora.inject_trainable_lora_extended(m, r=8)
        gen = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for t in lora.lora_parameters(m):
                t.copy_(torch.randn(t.shape, generator=gen) * 0.05)
        m.eval()  # freshly injected leaves are born in training mode (dropout 0.1)
    return m, g


def test_zero_up_is_identity_and_state_dict_round_trip(tmp_path):
    g = load("unet_tiny")
    m = UNetModel(**tiny_unet_params()).eval()
    m.load_state_dict(synth_state_dict(manifest("unet_tiny")), strict=True)
    lora.inject_trainable_lora_extended(m, r=8)  # lora_up starts at zero
    m.eval()
    with torch.no_grad():
        y = m(g["x"], g["ts"], context=g["ctx"], fps=16, timestep_cond=g["tc"])
    assert rel_l2(y, g["y"]) < 1e-5
    m2, _ = _tiny()
    lora.save_lora_weight(m2, str(tmp_path / "unet_lora.pt"))
    flat = torch.load(str(tmp_path / "unet_lora.pt"))
    assert len(flat) == 2 * sum(isinstance(x, lora._INJECTED) for x in m2.modules())
    m3 = UNetModel(**tiny_unet_params()).eval()
    m3.load_state_dict(synth_state_dict(manifest("unet_tiny")), strict=True)
    lora.inject_trainable_lora_extended(m3, r=8, loras=str(tmp_path / "unet_lora.pt"))
    for a, b in zip(lora.lora_parameters(m2), lora.lora_parameters(m3)):
        assert torch.equal(a, b)


def test_native_path_refuses_active_dropout():
    import pytest
    m, g = _tiny()
    m.train()
    with pytest.raises(RuntimeError):
        UNetEngine(m, EmuOps())(g["x"], g["ts"], g["ctx"], 16, g["tc"], None)


def test_collapse_remove_and_native_merge_agree():
    m, g = _tiny()
    args = (g["x"], g["ts"])
    kw = dict(context=g["ctx"], fps=16, timestep_cond=g["tc"])
    with torch.no_grad():
        y_branch = m(*args, **kw)  # eval mode: dropout off, branch active
        y_native = UNetEngine(m, EmuOps())(g["x"], g["ts"], g["ctx"], 16, g["tc"], None)
    assert rel_l2(y_branch, g["y"]) > 1e-3  # the LoRA delta is not a no-op
    assert rel_l2(y_native, y_branch) < 2e-5  # engine merges W + scale*up@down while packing
    lora.collapse_lora(m)
    lora.monkeypatch_remove_lora(m)
    assert not any(isinstance(x, lora._INJECTED) for x in m.modules())
    assert all(type(x) in (nn.Linear, nn.Conv2d, nn.Conv3d, nn.Conv1d) for x in m.modules()
               if isinstance(x, (nn.Linear, nn.Conv2d, nn.Conv3d, nn.Conv1d)))
    with torch.no_grad():
        y_collapsed = m(*args, **kw)
    assert rel_l2(y_collapsed, y_branch) < 2e-5
