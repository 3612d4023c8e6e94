"""FlatAdamW (CPU arithmetic) == torch.optim.AdamW step for step; parameters stay views of the flat buffer."""
import torch

from t2v_turbo_amd.dist import FlatGradSync
from t2v_turbo_amd.optim import FlatAdamW, update_ema_flat


def _run(device):
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(7, 5, device=device)), torch.nn.Parameter(torch.randn(33, device=device))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    sync = FlatGradSync(ps)
    opt = FlatAdamW(ps, sync, lr=1e-2, weight_decay=0.05)
    ropt = torch.optim.AdamW(ref, lr=1e-2, weight_decay=0.05)
    for it in range(4):
        gs = [torch.randn_like(p) * (it + 1) for p in ps]
        sync.zero_()
        for p, r, g in zip(ps, ref, gs):
            p.grad.copy_(g)
            r.grad = g.clone()
        torch.nn.utils.clip_grad_norm_(ref, 1.0)
        ropt.step()
        norm = opt.step(max_grad_norm=1.0)
        total = torch.sqrt(sum((g ** 2).sum() for g in gs))
        assert abs(float(norm) - float(total)) < 1e-4 * float(total)
    for p, r in zip(ps, ref):
        assert torch.allclose(p.detach(), r.detach(), rtol=2e-5, atol=2e-6)
        assert p.data_ptr() >= opt.flat_param.data_ptr()
    return opt


def test_flat_adamw_matches_torch_on_cpu():
    _run("cpu")
    t, s = torch.ones(10), torch.zeros(10)
    update_ema_flat(t, s, 0.9)
    assert torch.allclose(t, torch.full((10,), 0.9))


def test_flat_step_invalidates_the_engines_weight_fingerprint():
    """The parameters are ``.data`` views of the flat buffer: updating the buffer in place (fused kernel or torch op) does not
    move their version counters, which the native engines' packed-weight cache is keyed on.  A step must change the
    fingerprint, or a later inference call would sample with the previous step's (LoRA-merged) weights."""
    from t2v_turbo_amd.engine import params_fingerprint
    lin = torch.nn.Linear(6, 4)
    ps = list(lin.parameters())
    sync = FlatGradSync(ps)
    opt = FlatAdamW(ps, sync, lr=1e-2)
    fp0 = params_fingerprint(lin)
    sync.flat.normal_()
    before = lin.weight.detach().clone()
    opt.step()
    assert not torch.equal(lin.weight.detach(), before)
    assert params_fingerprint(lin) != fp0
