"""world_size-2 gloo: the data-parallel exchange of the distillation step (flat LoRA-gradient
all-reduce + scalar gather) reproduces single-process gradients of the mean loss."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.synth import synth_state_dict
from t2v_turbo_amd import cd_math, lora
from t2v_turbo_amd.dist import FlatGradSync, broadcast_parameters, gather_scalars
from t2v_turbo_amd.unet3d import UNetModel
from tests.util import manifest, tiny_unet_params


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _student():
    torch.manual_seed(0)
    m = UNetModel(**tiny_unet_params())
    m.load_state_dict(synth_state_dict(manifest("unet_tiny")), strict=True)
    m.requires_grad_(False)
    lora.inject_trainable_lora_extended(m, r=4)
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for t in lora.lora_parameters(m):
            t.copy_(torch.randn(t.shape, generator=gen) * 0.05)
    return m.eval()  # eval: dropout off so ranks and the single-process run are comparable


def _sample(i):
    g = torch.Generator().manual_seed(100 + i)
    return (torch.randn(1, 4, 2, 8, 8, generator=g), torch.tensor([999 - 240 * i]), torch.randn(1, 77, 128, generator=g),
            torch.randn(1, 256, generator=g), torch.randn(1, 4, 2, 8, 8, generator=g))


def _loss(m, i, l2=False):
    x, ts, ctx, tc, target = _sample(i)
    pred = m(x, ts, context=ctx, fps=16, timestep_cond=tc)
    if l2:  # (the Huber gradient amplifies 1e-6 forward differences between two implementations of the same network)
        return torch.nn.functional.mse_loss(pred, target)
    return cd_math.huber_loss(pred, target)


def _worker(rank, world, port, out, native=False, overlap="1"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), T2V_ASYNC_ALLREDUCE=overlap)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    m = _student()
    if rank == 1:  # perturb, then rank 0's weights must win
        with torch.no_grad():
            lora.lora_parameters(m)[0].add_(1.0)
    broadcast_parameters(m)
    sync = FlatGradSync(lora.lora_parameters(m))
    sync.zero_()
    if native:  # the student's forward / backward on the native gradient engine's dataflow (emulated ops), one engine per rank
        from tests.emu_ops import ReplayOps   # the record / replay protocol of the native backend: the all-reduce markers are
        m._native_ops_factory = ReplayOps     # entries of the recorded backward list, re-issued between its launches
        m.native_mode = "train"
    loss = _loss(m, rank, l2=native)
    loss.backward()
    info = {}
    if native:   # did the engine exchange its gradient arena itself, in how many pieces, and what did it leave for all_reduce_mean?
        eng = m.native_train_engine()
        info = {"segments": len(eng._handles), "rest": None if sync._rest_idx is None else int(sync._rest_idx.numel()),
                "cond": sum(p.numel() for p in eng.conditioning_parameters()), "arena": int(eng.e_used)}
    sync.all_reduce_mean()
    norm = sync.clip_grad_norm_(1e9)
    losses = gather_scalars(loss, loss * 2, loss * 0)
    if rank == 0:
        torch.save({"flat": sync.flat.clone(), "norm": norm, "losses": losses, "info": info}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_single_process(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    m = _student()
    params = lora.lora_parameters(m)
    l0, l1 = _loss(m, 0), _loss(m, 1)
    ((l0 + l1) / 2).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in params])
    assert got["flat"].numel() == ref.numel()
    assert float((got["flat"] - ref).norm() / ref.norm()) < 1e-5
    assert abs(float(got["norm"]) - float(ref.norm())) < 1e-4 * float(ref.norm())
    assert got["losses"].shape == (2, 3)
    assert abs(float(got["losses"][0, 0]) - float(l0)) < 1e-6 and abs(float(got["losses"][1, 0]) - float(l1)) < 1e-6


def test_two_rank_native_student_allreduce_matches_single_process(tmp_path):
    """Same exchange with each rank's student on the native gradient engine (``native_mode = "train"``): the engine's LoRA
    gradients land in the flat buffer through autograd, one all-reduce averages them."""
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, True), nprocs=2, join=True)
    got = torch.load(out)
    m = _student()
    m.native_mode = "off"
    params = lora.lora_parameters(m)
    l0, l1 = _loss(m, 0, l2=True), _loss(m, 1, l2=True)
    ((l0 + l1) / 2).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in params])
    assert float((got["flat"] - ref).norm() / ref.norm()) < 2e-4
    assert abs(float(got["losses"][0, 0]) - float(l0)) < 1e-5 and abs(float(got["losses"][1, 0]) - float(l1)) < 1e-5
    # the exchange was the overlapped one: the gradient arena all-reduced in segments from inside the backward, the conditioning
    # branch's tensors (and only those) left for all_reduce_mean
    info = got["info"]
    assert 2 <= info["segments"] <= 8 and info["rest"] == info["cond"] > 0, info
    # ... and it hands over what the single blocking all-reduce after the backward does
    out2 = str(tmp_path / "r0_blocking.pt")
    mp.spawn(_worker, args=(2, _free_port(), out2, True, "0"), nprocs=2, join=True)
    got2 = torch.load(out2)
    assert got2["info"]["segments"] == 0 and got2["info"]["rest"] is None
    assert float((got["flat"] - got2["flat"]).norm() / got2["flat"].norm()) < 1e-6


# ---- full fine-tuning (row a20) at world size 2 ----------------------------------------------------------------------------------------
def _full_student():
    from tests.test_unet_full_grad_cpu import _student as make
    return make()      # every parameter trainable, no LoRA (train_latent_t2v_turbo_v2.py:669,798-816)


def _full_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    m = _full_student()
    if rank == 1:  # perturb, then rank 0's weights must win
        with torch.no_grad():
            next(m.parameters()).add_(1.0)
    broadcast_parameters(m)
    params = list(m.parameters())
    sync = FlatGradSync(params)
    sync.zero_()
    from tests.emu_ops import EmuOps
    m._native_ops_factory = EmuOps     # the native gradient engine's dataflow with base-weight gradients (engine_full.py), one engine per rank
    m.native_mode = "train"
    loss = _loss(m, rank, l2=True)
    loss.backward()
    assert m._engine_box.full is not None and m._engine_box.full.training_full, "the full fine-tuning route was not taken"
    sync.all_reduce_mean()
    if rank == 0:
        torch.save({"flat": sync.flat.clone(), "loss": loss.detach()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_full_fine_tuning_allreduce_matches_single_process(tmp_path):
    """Full fine-tuning data-parallel over two ranks: each rank's forward / backward of EVERY UNet parameter on the native gradient
    engine (its gradients reach torch as views of one arena copy), ONE flat all-reduce averages them — against single-process autograd
    of the mean loss through the torch module."""
    out = str(tmp_path / "r0.pt")
    mp.spawn(_full_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    m = _full_student()
    m.native_mode = "off"
    params = list(m.parameters())
    l0, l1 = _loss(m, 0, l2=True), _loss(m, 1, l2=True)
    ((l0 + l1) / 2).backward()
    ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    assert got["flat"].numel() == ref.numel()
    assert float((got["flat"] - ref).norm() / ref.norm()) < 3e-4
    assert abs(float(got["loss"]) - float(l0.detach())) < 1e-5
