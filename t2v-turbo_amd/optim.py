"""Flat-buffer AdamW / gradient clip / EMA: the LoRA tensors, their gradients
(``dist.FlatGradSync``) and the optimizer moments each live in ONE contiguous fp32 buffer, so an optimizer
step is one fused HIP kernel (``t2v_adamw_step``) instead of bitsandbytes' CUDA-only 8-bit AdamW
(train_t2v_turbo_v1_lora.py:765-803) or 1150 small torch launches; the clip coefficient of
``clip_grad_norm_`` (:1193) is folded into the same pass.  On CPU tensors the same arithmetic runs in torch."""
import torch


class FlatAdamW:
    def __init__(self, params, grad_sync, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.params = [p for p in params if p.requires_grad]
        self.sync = grad_sync
        assert sum(p.numel() for p in self.params) == grad_sync.numel
        assert len(self.params) == len(grad_sync.params) and all(a is b for a, b in zip(self.params, grad_sync.params)), \
            "FlatAdamW and FlatGradSync must be built from the same parameter list, in the same order"
        dev = self.params[0].device
        self.flat_param = torch.empty(grad_sync.numel, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in self.params:  # parameters become views of the flat buffer
                self.flat_param[off:off + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[off:off + p.numel()].view_as(p)
                off += p.numel()
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self._ops = None
        self._ws = None
        self.after_step_hooks = []   # callables run after every step (engines that cache operand packs of the parameters)

    def _hip(self):
        if self._ops is None:
            self._ops = _shared_ops()
            self._ws = torch.empty(1025, dtype=torch.float32, device=self.flat_param.device)
        return self._ops

    @torch.no_grad()
    def grad_norm(self):
        g = self.sync.flat
        if g.is_cuda:
            ops = self._hip()
            ops.sumsq(g, self._ws[:1024], self._ws[1024:])
            return self._ws[1024].sqrt()
        return g.norm(2)

    @torch.no_grad()
    def step(self, max_grad_norm=None):
        """One AdamW step on the (already all-reduced) flat gradient; returns the pre-clip gradient norm."""
        self.step_count += 1
        g = self.sync.flat
        norm = self.grad_norm() if max_grad_norm is not None else None
        scale = 1.0
        if max_grad_norm is not None:
            scale = float(torch.clamp(max_grad_norm / (norm + 1e-6), max=1.0))
        b1, b2 = self.betas
        if g.is_cuda:
            self._hip().adamw_step(self.flat_param, g, self.exp_avg, self.exp_avg_sq, self.lr, b1, b2, self.eps,
                                   self.weight_decay, self.step_count, scale)
        else:
            gr = g * scale
            self.flat_param.mul_(1 - self.lr * self.weight_decay)
            self.exp_avg.mul_(b1).add_(gr, alpha=1 - b1)
            self.exp_avg_sq.mul_(b2).addcmul_(gr, gr, value=1 - b2)
            bc1, bc2 = 1 - b1 ** self.step_count, 1 - b2 ** self.step_count
            self.flat_param.addcdiv_(self.exp_avg, self.exp_avg_sq.sqrt() / bc2 ** 0.5 + self.eps, value=-self.lr / bc1)
        # The parameters are views of flat_param through ``.data``: neither the fused kernel nor an in-place op on the flat
        # buffer moves THEIR version counters, and the native engines key their packed (LoRA-merged) weights on those
        # (engine.params_fingerprint).  Touch one parameter so that a later inference call re-packs instead of sampling with
        # the weights of the previous step.
        self.params[0].add_(0.0)
        for hook in self.after_step_hooks:   # e.g. UNetGradEngine.invalidate_lora_packs (engine_lora.py)
            hook()
        return norm

    def zero_grad(self):
        self.sync.zero_()


_OPS = None


def _shared_ops():
    """One ``HipOps`` per process for the flat-buffer kernels (its constructor parses the GEMM tune table)."""
    global _OPS
    if _OPS is None:
        from .native import HipOps
        _OPS = HipOps()
    return _OPS


@torch.no_grad()
def update_ema_flat(target_flat, source_flat, rate=0.99, target_params=None):
    """EMA of a flat fp32 parameter buffer (utils/common_utils.py:307-319) in one kernel.  ``target_params``: the target
    network's parameters if they are ``.data`` views of ``target_flat`` — one of their version counters is moved so that the
    native engines re-pack the target's weights (see FlatAdamW.step)."""
    if target_flat.is_cuda:
        _shared_ops().ema_update(target_flat, source_flat, rate)
    else:
        target_flat.mul_(rate).add_(source_flat, alpha=1 - rate)
    if target_params:
        target_params[0].add_(0.0)
