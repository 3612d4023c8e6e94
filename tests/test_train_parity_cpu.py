"""CPU dry runs (emulated op backend, fp32) of the bodies of tests/test_gpu_train_parity.py: the trainer's route through
``unet(...)`` -> ``_NativeStudent`` -> ``loss.backward()`` with a FlatAdamW update between two steps, and the train-mode mask
replay with the masks REGENERATED from the recorded site geometry (what the GPU test has to do: the device keeps no masks)."""
import torch

from tests.emu_ops import EmuOps
from tests.test_gpu_train_parity import run_student_train_mode_vs_reference_oracle, run_train_mode_with_replayed_masks, run_trainer_route


def test_trainer_route_two_steps_cpu():
    run_trainer_route(torch.device("cpu"), lambda: EmuOps(strict=True), 1e-4, 0.9999)


def test_train_mode_regenerated_masks_cpu():
    run_train_mode_with_replayed_masks("cpu", EmuOps(strict=True), 2e-5, 1e-4, 0.99999, 1e-3)


def test_lora_grad_oracle_reproduces_the_reference_fixture():
    """oracle/lora_grad_oracle.student_reference (the checker of the GPU training-parity tests and of bench.py's distillation
    parity gate) against gradients the reference itself computed: tests/golden/unet_tiny_lora_grad.npz."""
    from oracle.lora_grad_oracle import per_tensor_agreement, student_reference
    from oracle.synth import synth_state_dict
    from t2v_turbo_amd import lora
    from t2v_turbo_amd.unet3d import UNetModel
    from tests.golden.make_golden_lora_grad import SEED_R, digests, draw_lora
    from tests.util import load, manifest, rel_l2, tiny_unet_params
    g, gg = load("unet_tiny"), load("unet_tiny_lora_grad")
    m = UNetModel(**tiny_unet_params()).eval()
    m.load_state_dict(synth_state_dict(manifest("unet_tiny")), strict=True)
    m.requires_grad_(False)
    lora.inject_trainable_lora_extended(m, r=64)
    draw_lora(lora.lora_parameters(m))
    r_out = torch.randn(g["x"].shape, generator=torch.Generator().manual_seed(SEED_R))
    y, dx, grads = student_reference(m.state_dict(), tiny_unet_params(), 64, g["x"], g["ts"], g["ctx"], 16, g["tc"], r_out)
    assert rel_l2(y, gg["out"]) < 2e-5 and rel_l2(dx, gg["dx"]) < 1e-4
    d, ref = torch.from_numpy(digests(grads)), gg["digests"]
    assert float(((d[:, 0] - ref[:, 0]).abs() / ref[:, 0]).max()) < 1e-4
    rows, zeros = per_tensor_agreement(grads, grads)
    assert not zeros and all(abs(c - 1) < 1e-12 and abs(q - 1) < 1e-12 for _, c, q in rows)


def test_lora_grad_oracle_reproduces_the_mid_width_reference_fixture():
    """The same at model_channels = 128 (tests/golden/unet_mid_lora_grad.npz, made by running the REFERENCE at that width:
    make_golden_lora_grad.py --width 128): the full-width training-parity chain compares the device with
    oracle/lora_grad_oracle.student_reference, whose module is this repository's own composite UNetModel — pinned to the reference at
    tiny width above, and here at a second width with 2 / 4 / 8 attention heads and 128-512-channel levels."""
    from oracle.lora_grad_oracle import student_reference
    from oracle.synth import manifest_of, synth_state_dict
    from t2v_turbo_amd import lora
    from t2v_turbo_amd.unet3d import UNetModel
    from tests.golden.make_golden_lora_grad import SEED_R, digests, draw_lora
    from tests.util import load, rel_l2, tiny_unet_params
    g, gg = load("unet_tiny"), load("unet_mid_lora_grad")
    cfg = tiny_unet_params(model_channels=int(gg["width"]))
    m = UNetModel(**cfg).eval()
    m.load_state_dict(synth_state_dict(manifest_of(m)), strict=True)
    m.requires_grad_(False)
    lora.inject_trainable_lora_extended(m, r=64)
    params = lora.lora_parameters(m)
    assert len(params) == 2 * int(gg["n_leaves"]) and [list(p.shape) for p in params] == [[int(v) for v in row[:p.dim()]] for row, p in zip(gg["shapes"], params)]
    draw_lora(params)
    r_out = torch.randn(g["x"].shape, generator=torch.Generator().manual_seed(SEED_R))
    y, dx, grads = student_reference(m.state_dict(), cfg, 64, g["x"], g["ts"], g["ctx"], 16, g["tc"], r_out)
    assert rel_l2(y, gg["out"]) < 2e-5 and rel_l2(dx, gg["dx"]) < 1e-4
    d, ref = torch.from_numpy(digests(grads)), gg["digests"]
    assert float(((d[:, 0] - ref[:, 0]).abs() / ref[:, 0]).max()) < 1e-4
    assert float(((d[:, 1:] - ref[:, 1:]).abs() / ref[:, :1]).max()) < 1e-3


def test_train_mode_masks_with_the_split_aware_lora_form_cpu():
    """``t2v_gemm_plan`` says the plain launch of a conv leaf would split K (the device library does for the 5x8 / 10x16 levels' long-K
    convs): the engine then keeps that leaf's LoRA branch as up-projection launches in front of a plain base leaf and the other leaves
    in the epilogue form (engine._lora_epilogue_pays, engine_unet_bwd.linear).  Same gate as the all-epilogue run above, on a backend
    whose plan splits every conv and every second linear launch."""
    from t2v_turbo_amd import native as nt

    class SplittingPlan(EmuOps):
        n_asked = n_split = 0

        def gemm_plan(self, a0, w, out, *, mode=nt.GEMM_LINEAR, **_):
            type(self).n_asked += 1
            split = mode != nt.GEMM_LINEAR or type(self).n_asked % 2 == 0
            type(self).n_split += int(split)
            return 0, (4 if split else 1)

    run_train_mode_with_replayed_masks("cpu", SplittingPlan(strict=True), 2e-5, 1e-4, 0.99999, 1e-3)
    assert SplittingPlan.n_split > 20 and SplittingPlan.n_asked > SplittingPlan.n_split


def test_student_train_mode_vs_reference_oracle_body_cpu():
    """Dry run of tests/test_gpu_train_parity.py::test_full_width_student_in_train_mode_with_replayed_masks at tiny width on the emulated
    backend: randomly initialised student, train mode, masks regenerated from the site geometry and patched into the module that
    oracle/lora_grad_oracle.student_reference builds from the state dict (its ``prepare`` hook)."""
    from tests.util import tiny_unet_params
    cfg = tiny_unet_params()
    run_student_train_mode_vs_reference_oracle(torch.device("cpu"), EmuOps(strict=True), cfg, (1, 4, 2, 8, 8), None, 20,
                                               2e-5, 1e-4, 0.99999, 1e-3, seed_model=11)
