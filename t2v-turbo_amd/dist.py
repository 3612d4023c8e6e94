"""Data-parallel plumbing for the distillation step: one process per GPU, full model replica,
ONE all-reduce(mean) of the trainable (LoRA) gradients per optimizer step over a flat contiguous
buffer (1150 tensors / 468.6 MB fp32 in v1), plus a 3-float all-gather for the
logged losses (C3).  ``backend="nccl"`` is RCCL on ROCm (xGMI); tests run the same code on gloo.

The reference gets this from accelerate -> DDP's 25 MB buckets; here the grads live in one buffer so
the collective is a single large message (the xGMI mesh is per-link bound: fewer, larger transfers)."""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None, single_process_group=False):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  The rank's GPU becomes the
    current device whenever CUDA is available (also under gloo: every native launch goes to the current device's stream).
    ``single_process_group``: create the group even at world size 1 (a one-rank RCCL communicator: smoke tests)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # the rank's GPU becomes the current device only when a launcher said which one it is (or there are several ranks): a
    # single process that chose its own device earlier (torch.cuda.set_device(1)) keeps it
    if torch.cuda.is_available() and local < torch.cuda.device_count() and ("LOCAL_RANK" in os.environ or world > 1):
        torch.cuda.set_device(local)
    if dist.is_initialized() or (world == 1 and not single_process_group):
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        kw["device_id"] = torch.device("cuda", local)
    if world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        kw.update(rank=0, world_size=1)
    dist.init_process_group(backend, **kw)


class FlatGradSync:
    """Views every trainable parameter's ``.grad`` into one flat buffer and averages it across ranks
    with a single all-reduce.  ``grad`` tensors stay views of the buffer, so optimisers see the
    averaged values without copies."""

    def __init__(self, params, dtype=torch.float32):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=dtype, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            p._t2v_flat_sync = self   # the native student writes its weight gradients straight into this buffer (unet3d._flat_grad_buffer)
            off += p.numel()
        self.numel = n
        self.force = False        # True: the native engine exchanges its gradients even in a one-rank group (RCCL smoke tests)
        self._rest_idx = None
        self._local_dirty = False

    def zero_(self):
        self.flat.zero_()
        self._rest_idx = None
        self._local_dirty = False

    def mark_engine_reduced(self, rest_idx):
        """The native gradient engine has written ALREADY AVERAGED gradients for its tensors (it all-reduces its gradient arena
        in segments while the backward runs: engine_lora); what is left for ``all_reduce_mean`` are the flat positions
        ``rest_idx`` (int64) — the conditioning branch's tensors, which torch differentiates after the engine's backward.

        INVARIANT: between ``zero_()`` and ``all_reduce_mean()`` nothing but that one overlapped backward may have written the
        engine's positions: an un-averaged local contribution there (a second student call of the same step that went through the
        torch composite path, say) would never be exchanged and the ranks would drift apart silently.  ``note_local_grads()`` is
        how such a writer says so; the next ``all_reduce_mean`` then exchanges the whole buffer.  (The averaged engine gradients are
        divided by the world size once more in that case — a step that mixes the two routes is refused instead: see below.)"""
        if getattr(self, "_local_dirty", False):
            raise RuntimeError("FlatGradSync: un-averaged local gradients were accumulated into the flat buffer in the same step as an "
                               "overlapped (already averaged) engine backward; exchange them first or use T2V_ASYNC_ALLREDUCE=0")
        self._rest_idx = rest_idx

    def note_local_grads(self):
        """A backward that leaves UN-averaged gradients in the flat buffer (torch autograd through the composite path, the engine's
        blocking mode) has run: the next exchange must cover the whole buffer."""
        if self._rest_idx is not None:
            raise RuntimeError("FlatGradSync: un-averaged local gradients after an overlapped (already averaged) engine backward in the "
                               "same step; use T2V_ASYNC_ALLREDUCE=0 for steps that mix the two routes")
        self._local_dirty = True

    def all_reduce_mean(self, async_op=False, force=False):
        """``force``: run the collective even in a one-rank group (smoke tests of the RCCL path)."""
        rest, self._rest_idx = self._rest_idx, None
        self._local_dirty = False
        if not dist.is_initialized() or (dist.get_world_size() == 1 and not (force or self.force)):
            return _DoneWork() if async_op else None
        if rest is not None:
            if rest.numel():
                buf = self.flat.index_select(0, rest)
                buf.div_(dist.get_world_size())
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
                self.flat.index_copy_(0, rest, buf)
            return _DoneWork() if async_op else None   # (the subset exchange is blocking: a caller that asked for a handle gets a completed one)
        self.flat.div_(dist.get_world_size())
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)

    def clip_grad_norm_(self, max_norm):
        """Global L2 clip on the (already averaged) flat buffer (accelerator.clip_grad_norm_)."""
        if self.flat.is_cuda and self.flat.dtype == torch.float32:  # deterministic two-pass sum of squares (t2v_sumsq), no ATen reduction
            from .optim import _shared_ops
            if getattr(self, "_norm_ws", None) is None:
                self._norm_ws = torch.empty(1025, dtype=torch.float32, device=self.flat.device)
            try:
                with torch.cuda.device(self.flat.device):
                    _shared_ops().sumsq(self.flat, self._norm_ws[:1024], self._norm_ws[1024:])
                norm = self._norm_ws[1024].sqrt()
            except Exception as e:  # noqa: BLE001 - the clip is exchange plumbing, not the hot path: say so and use torch's norm
                import warnings
                warnings.warn(f"FlatGradSync.clip_grad_norm_: native t2v_sumsq unavailable ({e}); using torch.norm", RuntimeWarning)
                norm = self.flat.norm(2)
        else:
            norm = self.flat.norm(2)
        scale = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
        self.flat.mul_(scale)
        return norm


class _DoneWork:
    """A completed work handle (``wait()`` / ``is_completed()``) for exchanges that had nothing asynchronous left to do."""

    def wait(self, *_a, **_k):
        return True

    def is_completed(self):
        return True


def broadcast_parameters(module, src=0):
    """Rank-0 weights to everyone (what the DDP constructor does, C1)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


def gather_scalars(*values):
    """All-gather a few 0-d losses in one message (C3) -> tensor [world, len(values)]."""
    v = torch.stack([torch.as_tensor(x, dtype=torch.float32).detach().reshape(()) for x in values])
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return v[None]
    dev = values[0].device if isinstance(values[0], torch.Tensor) else "cpu"
    v = v.to(dev)
    out = [torch.empty_like(v) for _ in range(dist.get_world_size())]
    dist.all_gather(out, v)
    return torch.stack(out)
