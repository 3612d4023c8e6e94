"""Native execution engines for the UNet forward and the VAE decode.

An engine walks the (reference-shaped) ``nn.Module`` tree once per input signature, packs the
leaf parameters into kernel layouts and *records* the sequence of C-ABI launches with all
buffers preallocated; later calls only refresh the small input buffers and replay the recorded
launches (optionally as one hipGraph).  Activations stay token-major ``[(b f)(h w), C]`` bf16 from
the first kernel to the last, so none of the reference's ``rearrange(...).contiguous()`` copies,
``repeat_interleave`` of context/embedding, ``torch.cat`` of skip connections or materialised
upsamples exist here.

The engine is written against an *ops backend* (``native.HipOps``); tests substitute a torch
emulation of the same interface to check the recorded dataflow on CPU.
"""
import collections
import math
import os

import torch
import torch.nn as nn

from . import native as nt
from .native import on_tensor_device
from .unet3d import (Downsample, ResBlock, SpatialTransformer, TemporalTransformer, TimestepEmbedSequential,
                     Upsample)


# =================================================================================== buffers
class BufferPool:
    """Size-bucketed device buffer reuse.  Launch order is fixed at record time, so a buffer released
    after its last consumer was recorded can be handed to a later producer (stream order = record order)."""

    def __init__(self, device):
        self.device = device
        self.free = collections.defaultdict(list)
        self.live = {}
        self.links = {}
        self.bytes = 0

    def get(self, rows, cols, dtype, zero=False):
        n = rows * cols
        nbytes = ((n * torch.empty((), dtype=dtype).element_size() + 255) // 256) * 256
        if self.free[nbytes]:
            base = self.free[nbytes].pop()
        else:
            base = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self.bytes += nbytes
        t = base.view(dtype)[:n].view(rows, cols)
        if zero:
            t.zero_()
        self.live[t.data_ptr()] = (nbytes, base)
        return t

    def link(self, t, side):
        """``side`` (another pool buffer: the statistics a producing GEMM wrote next to ``t``) is released together with ``t``."""
        self.links.setdefault(t.data_ptr(), []).append(side)

    def put(self, *tensors):
        for t in tensors:
            if t is not None:
                self.release_ptr(t.data_ptr())

    def release_ptr(self, ptr):
        if ptr not in self.live:        # (a linked side buffer may already have gone with its owner)
            return
        nbytes, base = self.live.pop(ptr)
        self.free[nbytes].append(base)
        for side in self.links.pop(ptr, ()):
            self.put(side)


class Act:
    """A token-major activation: one tensor, or two channel-concatenated parts (virtual concat)."""

    def __init__(self, parts, n_img, h, w, cs=None):
        self.parts = parts if isinstance(parts, (list, tuple)) else [parts]
        self.n_img, self.h, self.w = n_img, h, w
        # per part: the column statistics [rows / 32, C, 2] its producing GEMM wrote (t2v_gemm colstat_out), or None
        self.cs = list(cs) if cs is not None else [None] * len(self.parts)

    @property
    def t(self):
        assert len(self.parts) == 1
        return self.parts[0]

    @property
    def C(self):
        return sum(p.shape[1] for p in self.parts)

    @property
    def M(self):
        return self.parts[0].shape[0]

    @property
    def p1(self):
        return self.parts[1] if len(self.parts) > 1 else None


# =================================================================================== weights
def is_lora_leaf(mod):
    """A LoraInjected{Linear,Conv2d,Conv3d} (utils/lora.py:19-230) by its children — dictionary lookups, not ``getattr``: a missing
    attribute on an nn.Module costs an exception, and this runs over every module of the UNet on every training-path call."""
    d = mod._modules
    return "lora_up" in d and "lora_down" in d and (d.get("linear") is not None or d.get("conv") is not None)


def effective_weight_bias(mod, merge=True):
    """(weight, bias) of a Linear/Conv leaf; LoRA-injected leaves (utils/lora.py:19-230 layout:
    .linear|.conv, .lora_down, .lora_up, .scale[, .selector]) are merged on the fly:
    W + scale * up @ diag(sel) @ down — what ``collapse_lora`` (utils/lora.py:793-830) would bake in.
    ``merge=False`` (the training engine, which runs the LoRA branch as its own GEMMs): the frozen base only."""
    base = getattr(mod, "linear", None) or getattr(mod, "conv", None)
    if base is not None and hasattr(mod, "lora_up") and hasattr(mod, "lora_down"):
        if not merge:
            return base.weight.detach(), base.bias
        w = base.weight.detach().float()
        up = mod.lora_up.weight.detach().float().flatten(1)
        down = mod.lora_down.weight.detach().float().flatten(1)
        sel = getattr(mod, "selector", None)
        if isinstance(sel, (nn.Linear, nn.Conv2d, nn.Conv3d)):
            up = up @ sel.weight.detach().float().flatten(1)
        delta = (up @ down).reshape(w.shape)
        return w + float(mod.scale) * delta, base.bias
    return mod.weight.detach(), mod.bias


def leaf_out_channels(mod):
    """Output channels of a Linear / Conv leaf (a LoRA-injected leaf has its frozen base's) without merging anything."""
    base = getattr(mod, "linear", None) or getattr(mod, "conv", None)
    if base is not None and hasattr(mod, "lora_up") and hasattr(mod, "lora_down"):
        return base.weight.shape[0]
    return mod.weight.shape[0]


class Packer:
    """Packs leaf parameters into kernel layouts; cached until any parameter changes."""

    def __init__(self, wdtype, device, merge_lora=True):
        self.wdtype, self.device = wdtype, device
        self.merge_lora = merge_lora
        self.cache = {}
        self.makers = {}

    def wb(self, mod):
        return effective_weight_bias(mod, self.merge_lora)

    def _memo(self, key, fn):
        if key not in self.cache:
            self.cache[key] = fn()
            self.makers[key] = fn
        return self.cache[key]

    def refresh(self, ops=None):
        """Re-make every pack from the CURRENT parameters into the tensors that are already there (full fine-tuning: the weights move
        every optimizer step, the recorded launch lists keep pointing at the same packs).  Entries are re-made in the order they were
        first made, so a pack derived from another cached pack (transposes, fragment packs) sees its refreshed source.  ``ops``: the
        op backend, for the packs the library re-makes itself (transposes of a refreshed pack, conv packs).
        (Measured and not kept: the backward-only packs issued behind the forward's launches — 139.1 / 135.9 vs 139.8 / 135.2 ms per
        step, profiles/r06_full_finetune_wgrad_affine_rework_ab.jsonl.)"""
        def put(old, new):
            if isinstance(old, torch.Tensor):
                if old.data_ptr() != new.data_ptr():
                    old.copy_(new)
            elif isinstance(old, (tuple, list)):
                for o, n in zip(old, new):
                    put(o, n)
        by_id = None
        done = set()
        for key, fn in list(self.makers.items()):
            kind = key[0]
            if kind == "full_idx":   # (index tables: no weights inside)
                continue
            alias = getattr(fn, "param", None)
            if alias is not None and isinstance(self.cache[key], torch.Tensor) and alias.data_ptr() == self.cache[key].data_ptr():
                continue                 # (an fp32 parameter on the device IS its pack: nothing to re-make)
            if kind in ("mat", "mat_t") and len(key) == 2:
                # the two most common packs straight from the parameter into the existing tensor: ONE cast-and-copy kernel instead of a
                # cast into a temporary plus a device-to-device copy (1 100 of the 1 500 packs of the full-width UNet)
                if by_id is None:
                    by_id = getattr(self, "_mods_by_id", None)
                mod = None if by_id is None else by_id.get(key[1])
                src = self.cache.get(("mat", key[1])) if kind == "mat_t" and ("mat", key[1]) in done else None
                if (src is not None and ops is not None and hasattr(ops, "transpose") and src.dtype == self.cache[key].dtype
                        and (src.dtype == torch.bfloat16 or not src.is_cuda)):   # (the library's transpose is bf16; the emulated backend takes any)
                    # the data-gradient pack = the forward pack transposed, bf16 -> bf16 by the library's tiled transpose (the strided
                    # fp32 -> bf16 copy ran at ~ 300 GB/s: 30 us per pack, 7 ms per full fine-tuning step)
                    ops.transpose(src, src.shape[0], src.shape[1], self.cache[key])
                    done.add(key)
                    continue
                if mod is not None:
                    w = self.wb(mod)[0].detach()
                    w2 = w.reshape(w.shape[0], -1)
                    self.cache[key].copy_(w2 if kind == "mat" else w2.t())
                    done.add(key)
                    continue
            if getattr(fn, "into", False) and isinstance(self.cache[key], torch.Tensor):
                fn(self.cache[key], ops)      # (a maker that writes straight into the existing pack)
            else:
                put(self.cache[key], fn())
            done.add(key)

    def f32(self, p):
        if p is None:
            return None
        def make():
            return p.detach().to(self.device, torch.float32).contiguous()
        make.param = p       # (``refresh`` skips the pack while it still IS the parameter's storage)
        return self._memo(("f32", id(p)), make)

    def bias(self, mod):
        b = self.wb(mod)[1]
        if b is None:
            return None
        def make():
            return b.detach().to(self.device, torch.float32).contiguous()
        make.param = b
        return self._memo(("bias", id(mod)), make)

    def _remember(self, mod):
        if not hasattr(self, "_mods_by_id"):
            self._mods_by_id = {}
        self._mods_by_id[id(mod)] = mod

    def mat(self, mod):
        """[N, K] row-major weight of a Linear / 1x1 conv / k=1 Conv1d."""
        self._remember(mod)
        def make():
            w = self.wb(mod)[0]
            return w.reshape(w.shape[0], -1).to(self.device, self.wdtype).contiguous()
        return self._memo(("mat", id(mod)), make)

    def conv(self, mod):
        """[N, taps*Cin], tap-major: Conv2d [N,C,3,3] -> (ky,kx,c); Conv3d [N,C,3,1,1] -> (kt,c)."""
        def make(out=None, ops=None):
            return self._conv_tap_major(mod, out, ops)
        make.into = True
        return self._memo(("conv", id(mod)), make)

    def _conv_tap_major(self, mod, out=None, ops=None):
        w = self.wb(mod)[0]
        if self._repacked(w, out, ops, 0):
            return out
        if w.dim() == 5:
            w = w[:, :, :, 0, 0].permute(0, 2, 1)
        else:
            w = w.permute(0, 2, 3, 1)
        return self._permuted_into(w, out)

    @staticmethod
    def _repacked(w, out, ops, kind):
        """``refresh`` on the device: the conv parameter -> its existing bf16 pack by the library's repack kernel (t2v_repack_conv_f32;
        kind 0 forward pack, 1 data-gradient pack).  False where that does not apply (first making, CPU tensors, a merged LoRA weight)."""
        if out is None or ops is None or not hasattr(ops, "repack_conv") or w.dtype != torch.float32 or not w.is_contiguous():
            return False
        if os.environ.get("T2V_REPACK_NATIVE", "1") != "1":     # (A/B switch: torch's permute / flip / cast chain)
            return False
        if w.dim() == 5 and (w.shape[3] != 1 or w.shape[4] != 1):
            return False
        n, c = w.shape[0], w.shape[1]
        taps = w.numel() // (n * c)
        if taps > 9 or tuple(out.shape) != ((n, taps * c) if kind == 0 else (c, taps * n)) or not out.is_contiguous():
            return False
        if out.is_cuda and out.dtype != torch.bfloat16:
            return False
        ops.repack_conv(w, out, kind)
        return True

    def _permuted_into(self, w, out=None):
        """The [rows, -1] pack of the permuted weight view ``w`` — into ``out`` where that is the existing pack (``refresh``: cast and
        permutation as ONE kernel straight into the pack, no temporary and no second copy)."""
        if out is not None and tuple(out.shape) == (w.shape[0], w[0].numel()) and out.is_contiguous():
            out.view(w.shape).copy_(w)
            return out
        return w.reshape(w.shape[0], -1).to(self.device, self.wdtype).contiguous()

    def conv_slab(self, mod):
        """Slab-major pack of a 3x3 conv for t2v_conv_halo: [N][C/32][9][32], rows zero-padded to whole weight stages (native.pack_conv_slab)."""
        def make():
            cached = self.cache.get(("conv", id(mod)))   # (no second, tap-major copy is left behind for a conv the halo kernel takes)
            return nt.pack_conv_slab(cached if cached is not None else self._conv_tap_major(mod))
        return self._memo(("conv_slab", id(mod)), make)

    def mat_t(self, mod):
        """[K, N]^T pack of a Linear / 1x1 conv: the weight of its data gradient (dx = dy @ W)."""
        self._remember(mod)
        def make():
            w = self.wb(mod)[0]
            return w.reshape(w.shape[0], -1).t().to(self.device, self.wdtype).contiguous()
        return self._memo(("mat_t", id(mod)), make)

    def conv_dgrad(self, mod):
        """3x3 conv data gradient as a 3x3 conv over dy: w'[ci][(ky',kx'), co] = w[co][ci][2-ky'][2-kx']."""
        def make(out=None, ops=None):
            w = self.wb(mod)[0]          # [co, ci, 3, 3]
            if self._repacked(w, out, ops, 1):
                return out
            wd = w.flip(2, 3).permute(1, 2, 3, 0)      # [ci, ky', kx', co]
            return self._permuted_into(wd, out)
        make.into = True
        return self._memo(("conv_dgrad", id(mod)), make)

    def small_conv_dgrad(self, mod, cin_pad, cout_pad=None):
        """fp32 [cout'][9][cin'] pack for the direct small-channel conv computing the data gradient of ``mod``:
        cout' = mod's input channels (optionally zero-padded rows), cin' = mod's output channels padded to cin_pad."""
        def make():
            w = self.wb(mod)[0].float()  # [co, ci, 3, 3]
            wd = w.flip(2, 3).permute(1, 2, 3, 0)      # [ci, ky', kx', co]
            if cin_pad > wd.shape[-1]:
                wd = torch.nn.functional.pad(wd, (0, cin_pad - wd.shape[-1]))
            if cout_pad and cout_pad > wd.shape[0]:
                wd = torch.nn.functional.pad(wd, (0, 0, 0, 0, 0, 0, 0, cout_pad - wd.shape[0]))
            return wd.reshape(wd.shape[0], -1).to(self.device).contiguous()
        return self._memo(("small_dgrad", id(mod), cin_pad, cout_pad), make)

    def lpr(self, w):
        """Fragment pack (native.pack_linear_pr) of an [N, K] pack this Packer made — a plain ``mat`` / ``cat_mats`` matrix or the 64-row
        [value | gate] interleave of ``geglu`` — for t2v_linear_pr; keyed by the source pack, which the cache keeps alive."""
        return self._memo(("lpr", id(w)), lambda: nt.pack_linear_pr(w))

    def cat_mats(self, mods, tag):
        return self._memo((tag,) + tuple(id(m) for m in mods),
                          lambda: torch.cat([self.mat(m) for m in mods], dim=0).contiguous())

    def geglu(self, proj):
        """GEGLU projection packed in 64-row groups [32 value rows | 32 gate rows] (T2V_ACT_GEGLU)."""
        def make():
            w, b = self.wb(proj)
            inner = w.shape[0] // 2
            assert inner % 32 == 0
            wv, wg = w[:inner].reshape(inner // 32, 32, -1), w[inner:].reshape(inner // 32, 32, -1)
            wp = torch.cat([wv, wg], dim=1).reshape(2 * inner, -1).to(self.device, self.wdtype).contiguous()
            bp = torch.cat([b[:inner].reshape(-1, 32), b[inner:].reshape(-1, 32)], dim=1).reshape(-1)
            return wp, bp.detach().to(self.device, torch.float32).contiguous()
        return self._memo(("geglu", id(proj)), make)

    def mat_lnf(self, mods, norm, tag):
        """LayerNorm folded into the Linear(s) that consume it (t2v_gemm lnf_*): (W' = cat(W) diag(gamma) in the weight dtype,
        s = row sums of the ROUNDED W' (fp32: what the matrix cores multiply the mean with), t = b + cat(W) beta (fp32))."""
        def make():
            w = torch.cat([self.wb(m)[0].detach().float().reshape(self.wb(m)[0].shape[0], -1) for m in mods], dim=0).to(self.device)
            b = torch.cat([(self.wb(m)[1].detach().float() if self.wb(m)[1] is not None else torch.zeros(self.wb(m)[0].shape[0]))
                           .to(self.device) for m in mods])
            gamma, beta = norm.weight.detach().float().to(self.device), norm.bias.detach().float().to(self.device)
            wp = (w * gamma[None, :]).to(self.wdtype).contiguous()
            return wp, wp.float().sum(dim=1).contiguous(), (b + w @ beta).contiguous()
        return self._memo((tag, id(norm)) + tuple(id(m) for m in mods), make)

    def geglu_lnf(self, proj, norm):
        """``geglu`` pack (64-row groups [32 value | 32 gate]) of the LayerNorm-folded GEGLU projection: (W', s, t)."""
        def make():
            wp, s_vec, t_vec = self.mat_lnf([proj], norm, "geglu_lnf_src")
            inner = wp.shape[0] // 2
            assert inner % 32 == 0

            def pack(v):
                a, g = v[:inner].reshape(inner // 32, 32, -1), v[inner:].reshape(inner // 32, 32, -1)
                return torch.cat([a, g], dim=1).reshape(2 * inner, -1)
            return pack(wp).contiguous(), pack(s_vec[:, None]).reshape(-1).contiguous(), pack(t_vec[:, None]).reshape(-1).contiguous()
        return self._memo(("geglu_lnf", id(proj), id(norm)), make)

    def ffn(self, ff, norm):
        """Packed operands of t2v_ffn_fused for FeedForward ``ff`` (GEGLU projection + output Linear) behind LayerNorm ``norm``."""
        def make():
            proj, lin = ff.net[0].proj, ff.net[2]
            (w1, b1), (w2, b2) = self.wb(proj), self.wb(lin)
            dev = self.device
            return nt.ffn_pack(w1.detach().to(dev), None if b1 is None else b1.detach().to(dev), w2.detach().to(dev),
                               None if b2 is None else b2.detach().to(dev), norm.weight.detach().to(dev), norm.bias.detach().to(dev),
                               self.wdtype)
        return self._memo(("ffn", id(ff), id(norm)), make)

    def small_conv(self, mod, cin_pad=None):
        """fp32 [cout][9][cin] for the direct small-Cin conv."""
        def make():
            w = self.wb(mod)[0].float().permute(0, 2, 3, 1)  # N,3,3,C
            if cin_pad and cin_pad > w.shape[-1]:
                w = torch.nn.functional.pad(w, (0, cin_pad - w.shape[-1]))
            return w.reshape(w.shape[0], -1).to(self.device).contiguous()
        return self._memo(("small", id(mod), cin_pad), make)


def params_fingerprint(module, skip=()):
    """Changes when any parameter is updated in place (version counter), re-homed (data pointer), added or removed."""
    from .nn_util import walk_parameters
    fp = 0
    for p in walk_parameters(module):
        if id(p) in skip:
            continue
        fp = (fp * 1000003 + p._version + (p.data_ptr() & 0xFFFFFFF)) & 0xFFFFFFFFFFFF
    return fp


# =================================================================================== base engine
class _Engine:
    def __init__(self, ops):
        self.ops = ops
        self.adt = ops.act_dtype
        self.plans = {}
        self.fingerprint = None
        self.use_graph = os.environ.get("T2V_HIP_GRAPH", "0") == "1"

    # ---- helpers bound to the current recording -------------------------------------------------
    def _begin(self, device):
        """Start a recording.  Every plan OWNS what its launch list points at (``_own``): the recorded arguments are raw device
        pointers, so the buffers must outlive the plan, not just the recording (a second input signature used to free the first
        plan's pool, and going back to the first one replayed into reused memory).  Packed weights are shared by the plans of
        one weight version (``_check_weights`` drops all plans when a parameter changes)."""
        self.device = device
        self.pool = BufferPool(device)
        pk = getattr(self, "pk", None)
        if not (self.plans and pk is not None and pk.device == device and pk.merge_lora and pk.wdtype == self.adt):
            self.pk = Packer(self.adt, device)
        self.keep = []

    def _own(self, plan):
        plan["owned"] = (self.pool, self.pk, self.keep)
        return plan

    # A plan pins its activation pool (GBs at 16x40x64) and its hipGraph: a service that sees many batch sizes / resolutions
    # would grow without bound.  Least-recently-used plans beyond this many are dropped (their pool and graph go with them).
    max_plans = int(os.environ.get("T2V_MAX_PLANS", "4"))

    def _keep_plan(self, key, plan):
        """Register / refresh ``plan`` as the most recently used one and evict beyond ``max_plans``."""
        self.plans.pop(key, None)
        self.plans[key] = plan          # dicts keep insertion order: the first key is the least recently used
        while len(self.plans) > max(1, self.max_plans):
            old = next(iter(self.plans))
            self.plans.pop(old)

    def buf(self, rows, cols, dtype=None, zero=False):
        return self.pool.get(rows, cols, dtype or self.adt, zero)

    # the normalise pass of a GroupNorm can also touch the weights of the conv that follows it (t2v_group_norm `prefetch`: streaming
    # loads toward the Infinity Cache).  Measured on MI355X, interleaved A/B on one box: 24.16 / 24.15 ms per UNet step with it,
    # 23.86 / 23.83 without — the GEMMs do not get faster (19.5 vs 19.45 ms: their weights are not what they wait for) and the
    # GroupNorm passes pay for the extra loads.  Opt-in (T2V_PREFETCH=1), kept as a measured negative result.
    prefetch_weights = os.environ.get("T2V_PREFETCH", "0") == "1"

    def gn(self, x, norm, units, rows_per_unit, silu, eps=None, then=None):
        """GroupNorm(+SiLU) of an Act (possibly a virtual concat) -> new single-part tensor.  ``then``: the packed weight tensor
        of the launch that consumes the result (prefetched toward the Infinity Cache by the normalise pass)."""
        pf = {"prefetch": then} if (then is not None and self.prefetch_weights and getattr(self.ops, "is_native", False)) else {}
        ops = self.ops
        G = norm.num_groups
        eps = norm.eps if eps is None else eps
        out = self.buf(x.M, x.C)
        if self.fuse_gn and all(c is not None for c in x.cs) and rows_per_unit % 32 == 0:
            # the producing GEMMs left per-slab column statistics: no statistics pass over the tensor
            ws = self.buf(1, max(ops.group_norm_cs_ws_floats(units, rows_per_unit, G), 1), torch.float32)
            ops.group_norm_cs(x.cs[0], x.cs[1] if len(x.cs) > 1 else None, x.parts[0], x.p1, units, rows_per_unit, eps,
                              self.pk.f32(norm.weight), self.pk.f32(norm.bias), silu, ws, out, G, **pf)
        else:
            ws = self.buf(1, max(ops.group_norm_ws_floats(units, rows_per_unit, G, x.C), 1), torch.float32)
            ops.group_norm(x.parts[0], x.p1, units, rows_per_unit, eps, self.pk.f32(norm.weight), self.pk.f32(norm.bias),
                           silu, ws, out, G, **pf)
        self.pool.put(ws)
        return out

    # Normalisation statistics as by-products of the producing GEMMs (t2v_gemm rowstat_out / colstat_out) and LayerNorm folded
    # into the consuming GEMM (lnf_*): set by the engines that use them (UNetEngine); T2V_FUSE_GN=0 / T2V_FOLD_LN=0 switch back
    # to the standalone statistics passes / LayerNorm launches.
    fuse_gn = False
    fold_ln = False
    conv_halo = os.environ.get("T2V_CONV_HALO", "1") == "1"   # 3x3 convs on t2v_conv_halo where it takes them (0: always t2v_gemm)
    # Measured on MI355X (profiles/r03_fuse_ab.csv): the fold pays where the consuming GEMM is as wide as the rows it normalises
    # (the text cross-attention's q: +1.2 / +1.9 / +2.6 us on the launch and +2.3 / +0.7 / +0.4 us on the producer against a
    # 13.8 / 8.7 / 8.3 us LayerNorm at the three levels) and LOSES on the wide consumers — q|k|v (+17 us at the 320-channel
    # level) and the GEGLU projection (+50 us): every N-tile of the consumer re-loads its rows' statistics and re-scales in its
    # epilogue, the serial tail of a workgroup.  T2V_FOLD_LN_WIDE=1 folds those too (for the record, not for speed).
    fold_ln_wide = os.environ.get("T2V_FOLD_LN_WIDE", "0") == "1"
    # BasicTransformerBlock's feed-forward (LayerNorm, GEGLU projection, output projection, residual) as ONE launch where the
    # kernel exists (csrc/ffn.hip: C = 320, the level whose 105 MB hidden activation costs the most).  Correct on MI355X and
    # SLOWER than the three launches it replaces (212-220 us against 186-195 us at M = 40960; profiles/r03_ffn_fused_pmc.csv:
    # the matrix cores are busy 29 % of the time, the GEGLU / register-shuffle VALU work and the waits do not overlap them with
    # one wave per SIMD, and two do not fit: 120 + 240 + 48 registers of operands per lane).  Opt-in: T2V_FUSE_FF=1.
    fuse_ff = False

    def _colstat_for(self, a0, w, out, **kw):
        """A column-statistics buffer for this launch's output if the launch can carry it (linked to ``out`` in the pool)."""
        M, N = kw["M"], kw["N"]
        if not self.fuse_gn or M % 32 or N % 2 or out.dtype != self.adt:
            return None
        cs = self.buf(M // 32, 2 * N, torch.float32)
        if not self.ops.gemm_fuse_supported(a0, w, out, colstat=cs, **kw):
            self.pool.put(cs)
            return None
        self.pool.link(out, cs)
        return cs

    def linear(self, a, mod, *, residual=None, act=nt.ACT_NONE, out_dtype=None, w=None, bias="auto", N=None, ln=None,
               want_cs=False, want_rs=False, lnf=None):
        """``ln``: (LayerNorm module, destination buffer) — LayerNorm(out) as a second output of the same launch (t2v_gemm ln_*
        fields; N == 320 only, see ``ln_fusable``).  ``want_cs`` / ``want_rs``: also write the column / row statistics of the
        output where the launch can (returned as ``self.last_cs`` / ``self.last_rs``, None otherwise).  ``lnf``: (row statistics
        of ``a``, eps, s) — ``w`` / ``bias`` are a LayerNorm-folded pack (``Packer.mat_lnf``) and ``a`` the un-normalised rows."""
        w = self.pk.mat(mod) if w is None else w
        bias = self.pk.bias(mod) if isinstance(bias, str) else bias
        N = w.shape[0] if N is None else N
        out = self.buf(a.shape[0], N // 2 if act == nt.ACT_GEGLU else N, out_dtype)
        if ln is not None:
            norm, dst = ln
            ln = (self.pk.f32(norm.weight), self.pk.f32(norm.bias), norm.eps, dst)
        kw = dict(M=a.shape[0], N=N, bias=bias, residual=residual, act=act, **({"ln": ln} if ln is not None else {}))
        self.last_cs = self.last_rs = None
        if lnf is not None:
            kw["lnf"] = lnf
            if not self.ops.gemm_fuse_supported(a, w, out, **kw):   # the caller falls back to LayerNorm + the plain pack
                self.pool.put(out)
                return None
        elif want_cs:
            self.last_cs = self._colstat_for(a, w, out, **kw)
            if self.last_cs is not None:
                kw["colstat"] = self.last_cs
        elif want_rs and self.fold_ln and N % 64 == 0 and out.dtype == self.adt:   # (t2v_gemm: ld_rowstat = N / 16 must be a multiple of 4)
            rs = self.buf(a.shape[0], N // 16, torch.float32)
            if self.ops.gemm_fuse_supported(a, w, out, rowstat=rs, **kw):
                self.pool.link(out, rs)
                self.last_rs = kw["rowstat"] = rs
            else:
                self.pool.put(rs)
        if self._lpr_takes(a, w, out, kw):
            self.ops.linear_pr(a, self.pk.lpr(w), out, **kw)
        else:
            self.ops.gemm(a, w, out, **kw)
        return out

    # Short-K linears on the panel-resident kernel (csrc/linear_pr.hip) where it measured faster than the tuned t2v_gemm tile
    # (profiles/r06_linear_pr_vs_gemm.csv): the GEGLU projection and the q | k | v / q | k launches of the 320- and 640-channel levels.
    # The N = C launches (to_out, proj_in / proj_out: one chunk per wave, HBM-bound) tie or lose there and stay on t2v_gemm.
    linear_pr = os.environ.get("T2V_LINEAR_PR", "1") == "1"
    linear_pr_min_n = {320: int(os.environ.get("T2V_LPR_MIN_N_320", "640")), 640: int(os.environ.get("T2V_LPR_MIN_N_640", "1920")),
                       512: int(os.environ.get("T2V_LPR_MIN_N_512", "1536"))}   # (512: the 8-head temporal transformer behind the entry conv)
    if os.environ.get("T2V_LPR_K512", "1") != "1":
        del linear_pr_min_n[512]

    def _lpr_takes(self, a, w, out, kw):
        if not self.linear_pr or any(k in kw for k in ("ln", "lnf", "colstat", "rowstat")) or not hasattr(self.ops, "linear_pr_supported"):
            return False
        K, N = a.shape[1], kw["N"]
        if K not in self.linear_pr_min_n or w.shape[1] != K or w.shape[0] != N or w.stride(0) != K:
            return False
        if kw.get("act") != nt.ACT_GEGLU and N < self.linear_pr_min_n[K]:
            return False
        return self.ops.linear_pr_supported(a, w, out, **kw) == 1

    # LayerNorm in the panel fill of the ONE Linear that consumes it (t2v_gemm_desc::ln_in, csrc/linear_pr.hip): the temporal blocks'
    # norm1 / norm2 -> q | k | v and every block's norm3 -> GEGLU projection at the 320- and 640-channel levels — no t2v_layernorm launch,
    # no normalised tensor.  T2V_LN_IN=0: the separate launch.
    ln_in_fill = os.environ.get("T2V_LN_IN", "1") == "1"

    def linear_ln_in(self, a, norm, *, w, bias, act=nt.ACT_NONE):
        """Linear(LayerNorm(a)) as ONE t2v_linear_pr launch, or None where that kernel does not take the launch (the caller normalises first)."""
        if not self.ln_in_fill or tuple(norm.normalized_shape) != (a.shape[1],) or not norm.elementwise_affine:
            return None
        N = w.shape[0]
        out = self.buf(a.shape[0], N // 2 if act == nt.ACT_GEGLU else N)
        kw = dict(M=a.shape[0], N=N, bias=bias, residual=None, act=act,
                  ln_in=(self.pk.f32(norm.weight), self.pk.f32(norm.bias), norm.eps))
        if not self._lpr_takes(a, w, out, kw):
            self.pool.put(out)
            return None
        self.ops.linear_pr(a, self.pk.lpr(w), out, **kw)
        return out

    # LayerNorm as a by-product of the GEMM that produces its input.  Opt-in (T2V_FUSE_LN=1): measured on MI355X it removes 30
    # launches and 0.7 ms of t2v_layernorm per UNet step but adds 0.36 ms to the producing GEMMs (a second 26 MB write in their
    # epilogue) — 23.95 vs 24.06 ms per step in a same-box A/B, i.e. the kernel time is a wash and only the boundaries are saved.
    fuse_ln = os.environ.get("T2V_FUSE_LN", "0") == "1"

    def ln_fusable(self, C, norm):
        """The 160x320 workgroup tile holds whole rows only at N = 320 (the full-resolution level: 60 of the 99 LayerNorms).  A
        non-native (emulated) backend takes the fused form at every width, so that the CPU suite covers the dataflow."""
        wide_ok = C == 320 or not getattr(self.ops, "is_native", False)
        return self.fuse_ln and wide_ok and tuple(norm.normalized_shape) == (C,) and norm.elementwise_affine

    def conv(self, x, mod, mode, *, frames=0, rowvec=None, rowvec_div=0, residual=None, out_dtype=None, w=None, bias="auto",
             want_cs=True, extra=None, fallback=None, frozen_pack=False):
        """3x3 / strided / upsampled / temporal conv of an Act (virtual concat allowed); ``w`` / ``bias`` override the
        module's packed forward weights (data-gradient convs pass the flipped / transposed pack and no bias).  With
        ``fuse_gn`` the launch also writes its output's column statistics for the GroupNorm that follows every conv of the UNet."""
        own_w = w is None
        bias = self.pk.bias(mod) if isinstance(bias, str) else bias
        N = leaf_out_channels(mod) if own_w else w.shape[0]
        if mode == nt.GEMM_CONV3X3_S2:
            ho, wo = (x.h - 1) // 2 + 1, (x.w - 1) // 2 + 1
        elif mode == nt.GEMM_CONV3X3_UP2:
            ho, wo = 2 * x.h, 2 * x.w
        else:
            ho, wo = x.h, x.w
        M = x.n_img * ho * wo
        out = self.buf(M, N, out_dtype)
        kw = dict(M=M, N=N, a1=x.p1, mode=mode, n_img=x.n_img, h=x.h, wd=x.w, frames=frames, bias=bias, rowvec=rowvec,
                  rowvec_div=rowvec_div, residual=residual)
        # the slab-major pack of the halo kernel, if this launch can go there at all (None: it cannot).  The tap-major pack of
        # t2v_gemm is only built when a launch really needs it: a conv the halo kernel takes holds its weights ONCE per plan.
        ws = self._halo_pack(x, mod, w, own_w, frozen_pack, N, mode, out)
        if ws is not None and not self.ops.conv_halo_supported(x.parts[0], ws, out, **kw):
            ws = None
        zf = None
        if extra is not None:
            if w is None:
                w = self.pk.conv(mod)
            if not self._lora_epilogue_pays(x, w, out, kw, ws is not None):
                # (the gradient engine's LoRA branch as up-projection launches whose sum z is this launch's residual: the base leaf
                # keeps its split-K plan / goes to the halo kernel)
                zf, kw["residual"] = fallback()
                extra = None
                if ws is not None and not self.ops.conv_halo_supported(x.parts[0], ws, out, **kw):   # (the residual is new)
                    ws = None
        # 3x3 convs go to the halo-slab kernel (csrc/conv_halo.hip: the activation tile and its halo stay in LDS across the nine
        # taps; 14-32 % faster than the tuned t2v_gemm tiles at the UNet's three upper levels,
        # profiles/r04_conv_halo_v3_static_schedule.csv) when the launch has the module's own weights and a plain epilogue
        if ws is not None and extra is None:
            cs = None
            if want_cs and self.fuse_gn and M % 32 == 0:
                cs = self.buf(M // 32, 2 * N, torch.float32)
                if not self.ops.conv_halo_supported(x.parts[0], ws, out, colstat=cs, **kw):
                    self.pool.put(cs)
                    cs = None
            if cs is not None:
                self.pool.link(out, cs)
                kw["colstat"] = cs
            self.ops.conv_halo(x.parts[0], ws, out, **kw)
            if zf is not None:
                self.pool.put(zf)
            return Act(out, x.n_img, ho, wo, cs=[cs])
        if w is None:
            w = self.pk.conv(mod)
        if extra is not None:   # (the gradient engine: the LoRA branch in this launch's epilogue; ``fallback`` builds it as a residual)
            if self.ops.gemm_fuse_supported(x.parts[0], w, out, **kw, **extra):
                kw.update(extra)
            else:
                zf, kw["residual"] = fallback()
        cs = self._colstat_for(x.parts[0], w, out, **kw) if want_cs else None
        if cs is not None:
            kw["colstat"] = cs
        self.ops.gemm(x.parts[0], w, out, **kw)
        if zf is not None:
            self.pool.put(zf)
        return Act(out, x.n_img, ho, wo, cs=[cs])

    # output widths the halo kernel's wave tiles cover without padding: 80-channel wave tiles (the UNet's 320 / 640 / 1280)
    def _halo_width_ok(self, N):
        return N % 80 == 0

    # conv modes the halo kernel takes: stride-1 3x3, and (round 5) the same over a nearest-x2 upsampled source (the Upsample convs of
    # the UNet's decoder half and of the VAE decoder: openaimodel3d.py:98-112, ae_modules.py:141-156); T2V_HALO_UP2=0: stride-1 only
    halo_modes = (nt.GEMM_CONV3X3, nt.GEMM_CONV3X3_UP2) if os.environ.get("T2V_HALO_UP2", "1") == "1" else (nt.GEMM_CONV3X3,)

    def _halo_pack(self, x, mod, w, own_w, frozen_pack, N, mode, out):
        """The slab-major weight pack (``native.pack_conv_slab``) if the launch meets the halo kernel's static conditions, else None.
        Checked BEFORE anything is packed: stride-1 3x3 (or nearest-x2 + 3x3), the module's own or a frozen caller-given pack, 64-channel parts (the
        kernel walks 32-channel sub-slabs of 64-aligned parts), bf16 in and out."""
        if not (self.conv_halo and (own_w or frozen_pack) and mode in self.halo_modes and self._halo_width_ok(N)
                and hasattr(self.ops, "conv_halo_supported") and out.dtype == self.adt and self.pk.wdtype == self.adt):
            return None
        if any(part.shape[1] % 64 for part in x.parts):
            return None
        if own_w:
            return self.pk.conv_slab(mod)
        if w.dtype != self.adt:
            return None
        # (``frozen_pack``: a caller-given pack of FROZEN weights — the data-gradient convs' flipped base weights, which live in the
        # Packer's cache — is repacked once; packs that are rewritten every step, like the LoRA groups', must not be cached here)
        return self.pk._memo(("conv_slab_of", id(w)), lambda: (w, nt.pack_conv_slab(w)))[1]

    # where a LoRA conv leaf does better WITHOUT the epilogue form (profiles/r04_student_gemm_shapes.csv): (a) the plain launch would
    # split K (2560 x 1280 x 23040: 453 us in one split vs 162 us in four), (b) a 3x3 conv over >= ``lora_halo_min_c`` input channels
    # that the halo kernel REALLY takes (40960 x 320 x 5760: 157 us fused vs 112 + 20 us as halo conv + up-projection) — asked of
    # t2v_conv_halo_supported, not assumed: a launch it refuses would end on plain t2v_gemm + up-projection launches, the slowest form
    lora_split_aware = os.environ.get("T2V_LORA_SPLIT_AWARE", "1") == "1"
    lora_halo_min_c = int(os.environ.get("T2V_LORA_HALO_MIN_C", "640"))

    def _lora_epilogue_pays(self, x, w, out, kw, halo_takes):
        ops = self.ops
        if self.lora_split_aware and hasattr(ops, "gemm_plan") and ops.gemm_plan(x.parts[0], w, out, **kw)[1] > 1:
            return False
        cin = sum(p.shape[1] for p in x.parts)
        if self.lora_halo_min_c and cin >= self.lora_halo_min_c and halo_takes:
            return False
        return True

    # ---- plan management --------------------------------------------------------------------------------
    def _check_weights(self, module, skip=()):
        fp = params_fingerprint(module, skip)
        if fp != self.fingerprint:
            self.plans.clear()
            self.fingerprint = fp

    def _run(self, plan):
        ops = self.ops
        if not getattr(ops, "is_native", False):
            plan["fn"]()
            return
        if self.use_graph and plan.get("graph") is not None:   # (use_graph switched off again: back to the plain loop)
            plan["graph"].replay()
            return
        if self.use_graph and plan["runs"] >= 1 and not plan.get("graph_failed"):
            try:
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                # (thread-local capture mode: a HIP call from ANOTHER thread — the watchdog of an RCCL process group polling its events —
                # neither invalidates the capture nor faults in that thread; found as a once-in-five-runs abort of the one-rank RCCL test)
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    ops.replay(plan["rec"], ops.stream())
                plan["graph"] = g
                g.replay()
                return
            except Exception as e:  # capture unsupported -> stay on plain replay, loudly
                plan["graph_failed"] = str(e)
                import warnings
                warnings.warn(f"hipGraph capture failed, replaying launches instead: {e}")
        ops.replay(plan["rec"], ops.stream())
        plan["runs"] += 1


# =================================================================================== UNet
class UNetEngine(_Engine):
    def __init__(self, model, ops):
        super().__init__(ops)
        self.model = model
        # the inference engine takes GroupNorm statistics from the producing GEMMs' epilogues and folds the LayerNorms into the
        # GEMMs that consume them (the gradient engine, a subclass, keeps the standalone kernels: its tape saves their inputs)
        if type(self) is UNetEngine:
            self.fuse_gn = os.environ.get("T2V_FUSE_GN", "1") == "1"
            self.fold_ln = os.environ.get("T2V_FOLD_LN", "1") == "1"
            self.fuse_ff = os.environ.get("T2V_FUSE_FF", "0") == "1"

    @on_tensor_device
    def __call__(self, x, timesteps, context, fps=16, timestep_cond=None, motion_cond=None):
        m = self.model
        assert x.dim() == 5 and context is not None
        self._check_weights(m)
        fps_is_int = isinstance(fps, int)
        key = (tuple(x.shape), x.dtype, tuple(context.shape), context.dtype, fps_is_int,
               None if timestep_cond is None else (tuple(timestep_cond.shape), timestep_cond.dtype),
               None if motion_cond is None else (tuple(motion_cond.shape), motion_cond.dtype), x.device)
        plan = self.plans.get(key)
        # (a root module in eval mode whose sub-blocks were put back in train mode has live dropouts too: they are either applied
        # — the TemporalConvBlock ones — or refused, never silently ignored.  The Dropout(p > 0) modules are listed once per weight
        # version, so the per-call check is a loop over ~10^2 flags, not a walk over the module tree.)
        drops = self._active_tconv_dropouts() if (m.training or self._any_live_dropout()) else {}
        if plan is None or plan["training"] != m.training or plan.get("drop_sig") != tuple(sorted(drops.values())):
            self.drop_ps = drops   # id(nn.Dropout) -> p of the TemporalConvBlock dropouts that are live (train-mode frozen network)
            plan = self._own(self._record(x, timesteps, context, fps, timestep_cond, motion_cond))
            plan["training"] = m.training
            plan["drop_sig"] = tuple(sorted(drops.values()))
            self._keep_plan(key, plan)
        else:
            self._keep_plan(key, plan)
            st = plan["static"]
            if "seed" in st:   # a fresh mask per call (the launch list reads the seed from device memory: replays follow it)
                st["seed"].fill_(self._next_seed())
            st["x"].copy_(x)
            st["ts"].copy_(timesteps)
            st["ctx"].copy_(context)
            if m.fps_cond:
                if fps_is_int:
                    st["fps"].fill_(fps)
                else:
                    st["fps"].copy_(fps)
            if timestep_cond is not None:
                st["tc"].copy_(timestep_cond)
            if motion_cond is not None:
                st["mc"].copy_(motion_cond)
            self._run(plan)
        self._publish_probs(plan)
        return plan["out"].clone()

    # ---- train-mode frozen network (the v1 distillation teacher) ----------------------------------------------------------
    # train_t2v_turbo_v1_lora.py never calls .eval() on its teacher (:621-626; calls at :1105-1134, under no_grad), so the
    # Dropout(p = 0.1) layers of every TemporalConvBlock (openaimodel3d.py:282-294) are LIVE in the reference's teacher forwards.
    # The inference dataflow applies them with the counter-based masks of t2v_dropout_bf16 between the GroupNorm + SiLU and the
    # (3,1,1) conv, exactly where the gradient engine puts them for the student; any other active Dropout still refuses.
    def _any_live_dropout(self):
        # every nn.Dropout is listed (not only p > 0: p can be raised in place with no parameter change); the list is rebuilt per weight
        # version and on every 64th call (a Dropout swapped into the tree without touching a parameter is then seen within 64 calls at
        # 1 ms per walk, instead of never)
        from .nn_util import walk_modules
        cache = getattr(self, "_dropout_mods", None)
        if cache is None or cache[0] != self.fingerprint or cache[2] >= 64:
            cache = self._dropout_mods = [self.fingerprint, [mod for mod in walk_modules(self.model) if isinstance(mod, nn.Dropout)], 0]
        cache[2] += 1
        return any(mod.training and mod.p > 0 for mod in cache[1])

    def _active_tconv_dropouts(self):
        from .nn_util import walk_modules
        from .unet3d import TemporalConvBlock
        mods = walk_modules(self.model)
        active = [mod for mod in mods if isinstance(mod, nn.Dropout) and mod.p > 0 and mod.training]
        if not active:
            return {}
        known = {}
        for blk in mods:
            if isinstance(blk, TemporalConvBlock):
                for stg in (blk.conv1, blk.conv2, blk.conv3, blk.conv4):
                    for layer in stg:
                        if isinstance(layer, nn.Dropout) and layer.p > 0 and layer.training:
                            known[id(layer)] = float(layer.p)
        other = [mod for mod in active if id(mod) not in known]
        if other:
            raise RuntimeError(f"native UNet path: {len(other)} Dropout(p>0) module(s) in training mode that the inference engine does "
                               "not apply (only the TemporalConvBlock dropouts of a frozen train-mode network are): call .eval() first")
        return known

    def _next_seed(self):
        """One mask seed per forward.  ``seed_source`` (tests: a list / iterator of ints) overrides torch's generator."""
        src = getattr(self, "seed_source", None)
        if src is not None:
            return int(next(src))
        return int(torch.randint(0, 2 ** 62, (1,)).item())

    def _publish_probs(self, plan):
        for attn, probs in plan["probs"]:
            attn.attention_probs = probs

    # ---- recording ----------------------------------------------------------------------------------------
    def _record(self, x, timesteps, context, fps, timestep_cond, motion_cond):
        m, ops = self.model, self.ops
        self._begin(x.device)
        B, Cin, F, H, W = x.shape
        self.B, self.F = B, F
        st = {"x": x.detach().clone().contiguous(), "ts": timesteps.detach().to(torch.int64).clone(),
              "ctx": context.detach().clone().contiguous()}
        if m.fps_cond:
            st["fps"] = (torch.full_like(st["ts"], fps) if isinstance(fps, int) else fps.detach().to(torch.int64).clone())
        if timestep_cond is not None:
            st["tc"] = timestep_cond.detach().clone().contiguous()
        if motion_cond is not None:
            st["mc"] = motion_cond.detach().clone().contiguous()
        if getattr(self, "drop_ps", None):
            st["seed"] = torch.full((1,), self._next_seed(), dtype=torch.int64, device=x.device)
        self.seed_t = st.get("seed")
        out = torch.empty_like(st["x"][:, :m.out_channels].contiguous()) if m.out_channels != Cin else torch.empty_like(st["x"])
        plan = {"static": st, "out": out, "probs": [], "runs": 0}
        self.plan = plan

        def body():
            self._forward(st, out)

        if getattr(ops, "is_native", False):
            ops.init()
            ops.recording = []
            try:
                body()
            finally:
                plan["rec"] = ops.recording
                ops.recording = None
        else:
            plan["fn"] = body
            body()
        plan["pool_bytes"] = self.pool.bytes
        return plan

    def _forward(self, st, out):
        m, ops, pk = self.model, self.ops, self.pk
        B, F = self.B, self.F
        self.drop_sites = []   # ([nn.Dropout], kind, geometry) per applied mask, in launch order (tests replay them inside the torch module)
        x = st["x"]
        _, Cin, _, H, W = x.shape
        mc = m.model_channels
        L, D = st["ctx"].shape[1], st["ctx"].shape[2]
        # ---- conditioning vectors (M = B rows; openaimodel3d.py:683-706) ----------------------------------
        t_emb = self.buf(B, mc)
        ops.timestep_embedding(st["ts"], mc, False, t_emb)
        emb_in = t_emb
        if "tc" in st:
            tcb = self.buf(B, st["tc"].shape[1])
            ops.cast(st["tc"], tcb)
            if "mc" in st:
                cond = self.linear(tcb, m.time_cond_proj)
                mcb = self.buf(B, st["mc"].shape[1])
                ops.cast(st["mc"], mcb)
                mproj = self.linear(mcb, m.motion_cond_proj)
                emb_in = self.buf(B, mc)
                ops.gemm(cond, pk.mat(m.combine_proj), emb_in, M=B, N=mc, a1=mproj, residual=t_emb)
            else:
                emb_in = self.linear(tcb, m.time_cond_proj, residual=t_emb)
        e1 = self.linear(emb_in, m.time_embed[0], act=nt.ACT_SILU)
        emb = self.linear(e1, m.time_embed[2])
        if m.fps_cond:
            f_emb = self.buf(B, mc)
            ops.timestep_embedding(st["fps"], mc, False, f_emb)
            f1 = self.linear(f_emb, m.fps_embedding[0], act=nt.ACT_SILU)
            emb = self.linear(f1, m.fps_embedding[2], residual=emb)
        emb_s = self.buf(B, emb.shape[1])
        ops.silu(emb, emb_s)
        # every ResBlock's emb_layers Linear in one GEMM (rows are identical across frames: K17)
        resblocks = [mod for mod in m.modules() if isinstance(mod, ResBlock)]
        self.emb_off, off = {}, 0
        for rb in resblocks:
            self.emb_off[id(rb)] = off
            off += rb.out_channels
        lins = [rb.emb_layers[1] for rb in resblocks]
        w_all = pk.cat_mats(lins, "emb_all")
        b_all = pk._memo(("emb_all_bias",) + tuple(id(l) for l in lins),
                         lambda: torch.cat([pk.bias(l) for l in lins]).contiguous())
        self.emb_all = self.linear(emb_s, None, w=w_all, bias=b_all, out_dtype=torch.float32)
        # ---- context in activation dtype, shared by all frames of a clip (K11) ----------------------------
        self.ctx = self.buf(B * L, D)
        ops.cast(st["ctx"], self.ctx)
        self.ctx_len = L
        self.ctx_kv = {}
        # ---- input conv on the 4-channel latent -----------------------------------------------------------------
        xt = self.buf(B * F * H * W, Cin)
        ops.ncfhw_to_tokens(x, xt)
        conv_in = m.input_blocks[0][0]
        h0 = self.buf(B * F * H * W, leaf_out_channels(conv_in))
        ops.conv_small(xt, B * F, H, W, pk.small_conv(conv_in), pk.bias(conv_in), h0)
        h = Act(h0, B * F, H, W)
        hs = []
        for i, block in enumerate(m.input_blocks):
            if i > 0:
                h = self.run_sequential(block, h)
            if i == 0 and m.addition_attention:
                h = self.run_sequential(m.init_attn, h)
            hs.append(h)
        h = self.run_sequential(m.middle_block, h)
        for block in m.output_blocks:
            skip = hs.pop()
            h = self.run_sequential(block, Act([h.t, skip.t], h.n_img, h.h, h.w, cs=[h.cs[0], skip.cs[0]]), release=[h.t, skip.t])
        # ---- out: GroupNorm -> SiLU -> conv to 4 channels, fp32, back to (b c f h w) ------------------------------
        t = self.gn(h, m.out[0], B * F, H * W, True, then=self.pk.conv(m.out[2]))
        y = self.conv(Act(t, h.n_img, h.h, h.w), m.out[2], nt.GEMM_CONV3X3, out_dtype=torch.float32)
        ops.tokens_to_ncfhw(y.t, out)

    def run_sequential(self, seq, h, release=()):
        assert isinstance(seq, TimestepEmbedSequential)
        first = True
        for layer in seq:
            if isinstance(layer, ResBlock):
                nh = self.res_block(layer, h)
            elif isinstance(layer, SpatialTransformer):
                nh = self.spatial_transformer(layer, h)
            elif isinstance(layer, TemporalTransformer):
                nh = self.temporal_transformer(layer, h)
            elif isinstance(layer, Downsample):
                assert layer.use_conv, "avg-pool downsample is not used by the VideoCrafter2 config"
                nh = self.conv(h, layer.op, nt.GEMM_CONV3X3_S2)
            elif isinstance(layer, Upsample):
                assert layer.use_conv
                nh = self.conv(h, layer.conv, nt.GEMM_CONV3X3_UP2)
            else:
                raise NotImplementedError(f"native path: unsupported layer {type(layer).__name__}")
            # intermediates inside a sequential die as soon as the next layer consumed them; the
            # sequential's *input* belongs to the caller (it may be a skip connection)
            if not first:
                self.pool.put(*h.parts)
            h, first = nh, False
        self.pool.put(*release)
        return h

    # ---- blocks ------------------------------------------------------------------------------------------------------
    def res_block(self, rb, x):
        B, F = self.B, self.F
        hw = x.h * x.w
        off, cout = self.emb_off[id(rb)], rb.out_channels
        t = self.gn(x, rb.in_layers[0], B * F, hw, True, then=self.pk.conv(rb.in_layers[2]))
        h1 = self.conv(Act(t, x.n_img, x.h, x.w), rb.in_layers[2], nt.GEMM_CONV3X3,
                       rowvec=self.emb_all[:, off:off + cout], rowvec_div=F * hw)
        self.pool.put(t)
        t2 = self.gn(h1, rb.out_layers[0], B * F, hw, True, then=self.pk.conv(rb.out_layers[3]))
        self.pool.put(h1.t)
        if isinstance(rb.skip_connection, nn.Identity):
            skip, own_skip = x.t, False
        else:
            sc = rb.skip_connection
            w = self.pk.mat(sc) if effective_weight_bias(sc)[0].shape[-1] == 1 else None
            assert w is not None, "3x3 skip convs (use_conv=True) are not built by the VideoCrafter2 config"
            skip = self.buf(x.M, cout)
            self.ops.gemm(x.parts[0], w, skip, M=x.M, N=cout, a1=x.p1, bias=self.pk.bias(sc))
            own_skip = True
        h2 = self.conv(Act(t2, x.n_img, x.h, x.w), rb.out_layers[3], nt.GEMM_CONV3X3, residual=skip)
        self.pool.put(t2)
        if own_skip:
            self.pool.put(skip)
        if rb.use_temporal_conv:
            h2 = self.temporal_conv_block(rb.temopral_conv, h2)
        return h2

    def temporal_conv_block(self, tc, h2):
        """4 x [GroupNorm over all frames -> SiLU -> (3,1,1) conv] + identity (openaimodel3d.py:257-309); consumes h2."""
        B, F = self.B, self.F
        hw = h2.h * h2.w
        y = h2
        for i, stage in enumerate((tc.conv1, tc.conv2, tc.conv3, tc.conv4)):
            tt = self.gn(y, stage[0], B, F * hw, True, then=self.pk.conv(stage[-1]))
            for layer in stage:   # train-mode frozen network: the stage's Dropout between SiLU and the conv (openaimodel3d.py:282-294)
                p_drop = getattr(self, "drop_ps", {}).get(id(layer)) if isinstance(layer, nn.Dropout) else None
                if p_drop:
                    site = len(self.drop_sites)
                    self.drop_sites.append(([layer], "tconv", (B, F, h2.h, h2.w)))
                    self.ops.dropout(tt, None, tt, tt.shape[1], p_drop, self.seed_t, site)
            ny = self.conv(Act(tt, h2.n_img, h2.h, h2.w), stage[-1], nt.GEMM_TCONV3, frames=F,
                           residual=h2.t if i == 3 else None)
            self.pool.put(tt)
            if y is not h2:
                self.pool.put(y.t)
            y = ny
        self.pool.put(h2.t)
        return y

    def context_kv(self, attn):
        """K and V^T of the text context for EVERY cross-attention layer in two GEMMs (once per clip, not
        per frame and not per layer: the 16 to_k / to_v weights are stacked along N, like emb_layers)."""
        if self.ctx_kv is None or not self.ctx_kv:
            ops, pk, m, B, L = self.ops, self.pk, self.model, self.B, self.ctx_len
            layers = [blk.attn2 for mod in m.modules() if isinstance(mod, SpatialTransformer)
                      for blk in mod.transformer_blocks]
            total = sum(a.heads * a.dim_head for a in layers)
            wk = pk.cat_mats([a.to_k for a in layers], "ctx_k_all")
            wv = pk.cat_mats([a.to_v for a in layers], "ctx_v_all")
            kp = ((L + 63) // 64) * 64
            k_all = self.buf(B * L, total)
            ops.gemm(self.ctx, wk, k_all, M=B * L, N=total)
            vt_all = self.buf(B * total, kp)
            if kp != L:
                ops.fill_zero(vt_all)
            ops.gemm(wv, self.ctx, vt_all, M=total, N=L, batch=B, w_strides=(L * self.ctx.stride(0), 0),
                     o_strides=(total * kp, 0))
            off = 0
            for a in layers:
                inner = a.heads * a.dim_head
                self.ctx_kv[id(a)] = (k_all[:, off:off + inner], vt_all[off:off + inner], kp, total * kp)
                off += inner
        return self.ctx_kv[id(attn)]

    def _check_heads(self, attn):
        if attn.dim_head != 64:
            raise nt.NativeError(f"native attention kernels need dim_head == 64 (got {attn.dim_head})")

    def transformer_block(self, blk, y, x_geom, temporal, ln1=None, rs=None):
        """BasicTransformerBlock on token rows y [M, C] (attention.py:300-311).  ``ln1``: norm1(y) if the GEMM that produced y
        already wrote it (the buffer then serves the block's other two LayerNorms as well).  ``rs``: the row statistics of y its
        producer wrote (``fold_ln``): a LayerNorm whose consumer is ONE GEMM over the normalised rows (temporal q|k|v, the text
        cross-attention's q, the GEGLU projection) is then folded into that GEMM — no LayerNorm launch, no normalised tensor."""
        ops, pk = self.ops, self.pk
        B, F = self.B, self.F
        M, C = y.shape
        n_img, hw = x_geom
        a1, a2 = blk.attn1, blk.attn2
        self._check_heads(a1)
        inner = a1.heads * a1.dim_head
        box = {"ln": ln1}
        fold = self.fold_ln and ln1 is None
        fuse2, fuse3 = (self.ln_fusable(C, blk.norm2) and not fold), (self.ln_fusable(C, blk.norm3) and not fold)

        def ln_buf():
            if box["ln"] is None:
                box["ln"] = self.buf(M, C)
            return box["ln"]

        def lnorm(norm, src):
            ops.layernorm(src, pk.f32(norm.weight), pk.f32(norm.bias), norm.eps, ln_buf())
            return box["ln"]

        def foldable(norm, stats):
            return (fold and stats is not None and tuple(norm.normalized_shape) == (C,) and norm.elementwise_affine and
                    C % 64 == 0 and C <= 1280)

        wide = fold and self.fold_ln_wide   # q|k|v and the GEGLU projection (see fold_ln_wide)

        def folded(src, norm, stats, pack, act=nt.ACT_NONE):
            """The GEMM that consumes LayerNorm(src), run on the raw rows with the LayerNorm folded in — or None where the fold
            does not apply or the launch cannot carry it (the caller then normalises first)."""
            if not foldable(norm, stats):
                return None
            wp, s_vec, t_vec = pack()
            return self.linear(src, None, w=wp, bias=t_vec, act=act, lnf=(stats, norm.eps, s_vec))

        def temporal_attn(attn, norm, src, stats):
            # q | k | v of LayerNorm(src) as ONE GEMM
            qkv = folded(src, norm, stats if wide else None, lambda: pk.mat_lnf([attn.to_q, attn.to_k, attn.to_v], norm, "qkv_lnf"))
            if qkv is None and src is not box["ln"]:
                qkv = self.linear_ln_in(src, norm, w=pk.cat_mats([attn.to_q, attn.to_k, attn.to_v], "qkv"), bias=None)
            if qkv is None:
                qkv = self.linear(lnorm(norm, src) if src is not box["ln"] else src, None,
                                  w=pk.cat_mats([attn.to_q, attn.to_k, attn.to_v], "qkv"), bias=None)
            o = self.buf(M, inner)
            probs = None
            if attn.record_attn_probs:
                probs = torch.empty(B * hw * attn.heads, F, F, dtype=torch.float32, device=self.device)
                self.plan["probs"].append((attn, probs))
            ops.attn_temporal(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], o, B, F, hw, attn.heads,
                              attn.scale, probs)
            self.pool.put(qkv)
            return o

        def spatial_self_attn(attn, src):
            qk = self.linear(src, None, w=pk.cat_mats([attn.to_q, attn.to_k], "qk"), bias=None)
            kp = ((hw + 63) // 64) * 64
            vt = self.buf(n_img * inner, kp)
            if kp != hw:  # padding keys get P = 0 in the kernel; their V^T columns only have to be finite
                ops.fill_zero(vt)
            # V^T[c, token] = Wv[c,:] . x[token,:]: weights as the row operand, tokens as the column operand
            ops.gemm(pk.mat(attn.to_v), src, vt, M=inner, N=hw, batch=n_img, w_strides=(hw * src.stride(0), 0),
                     o_strides=(inner * kp, 0))
            o = self.buf(M, inner)
            ops.attn_spatial(qk[:, :inner], qk[:, inner:], vt, kp, o, n_img, hw, hw, attn.heads, 1, attn.scale)
            self.pool.put(qk, vt)
            return o

        def cross_attn(attn, norm, src, stats):
            q = folded(src, norm, stats, lambda: pk.mat_lnf([attn.to_q], norm, "q_lnf"))
            if q is None:
                q = self.linear(lnorm(norm, src) if src is not box["ln"] else src, attn.to_q, bias=None)
            k, vt, kp, vt_stride = self.context_kv(attn)
            o = self.buf(M, inner)
            ops.attn_spatial(q, k, vt, kp, o, n_img, hw, self.ctx_len, attn.heads, F, attn.scale, vt_stride)
            self.pool.put(q)
            return o

        # attn1: self attention (spatial or temporal).  Every consumer of a LayerNorm output has been launched before the next
        # producer overwrites the shared buffer (stream order), fused or not.
        if temporal:
            o = temporal_attn(a1, blk.norm1, ln1 if ln1 is not None else y, rs)
        else:  # two consumers (q|k and V^T, the latter with the tokens as its column operand): the LayerNorm stays a launch
            o = spatial_self_attn(a1, ln1 if ln1 is not None else lnorm(blk.norm1, y))
        # row statistics only where the next LayerNorm will be folded: the spatial block's norm2 (-> cross-attention q)
        y1 = self.linear(o, a1.to_out[0], residual=y, ln=(blk.norm2, ln_buf()) if fuse2 else None, want_rs=fold and (wide or not temporal))
        rs1 = self.last_rs
        self.pool.put(o)
        # attn2: temporal self attention again, or text cross attention
        src = box["ln"] if fuse2 else y1
        o = temporal_attn(a2, blk.norm2, src, rs1) if temporal else cross_attn(a2, blk.norm2, src, rs1)
        y2 = self.linear(o, a2.to_out[0], residual=y1, ln=(blk.norm3, ln_buf()) if fuse3 else None, want_rs=wide)
        rs2 = self.last_rs
        self.pool.put(o, y1)
        # GEGLU feed-forward
        proj = blk.ff.net[0]
        assert hasattr(proj, "proj"), "non-gated FeedForward is not built by the VideoCrafter2 config"
        if (self.fuse_ff and ops.ffn_fused_supported(C) and tuple(blk.norm3.normalized_shape) == (C,) and blk.norm3.elementwise_affine
                and not fuse3 and leaf_out_channels(blk.ff.net[2]) == C and leaf_out_channels(proj.proj) == 8 * C):
            # LayerNorm -> GEGLU projection -> output projection -> + residual as ONE launch (csrc/ffn.hip): the 4C-wide hidden
            # activation stays in registers
            w1p, b1p, w2p, b2f = pk.ffn(blk.ff, blk.norm3)
            y3 = self.buf(M, C)
            ops.ffn_fused(y2, w1p, b1p, w2p, b2f, blk.norm3.eps, y3)
            self.pool.put(y2, box["ln"])
            return y3
        g = folded(y2, blk.norm3, rs2 if wide else None, lambda: pk.geglu_lnf(proj.proj, blk.norm3), act=nt.ACT_GEGLU)
        if g is None and not fuse3:
            wg, bg = pk.geglu(proj.proj)
            g = self.linear_ln_in(y2, blk.norm3, w=wg, bias=bg, act=nt.ACT_GEGLU)
        if g is None:
            src = box["ln"] if fuse3 else lnorm(blk.norm3, y2)
            wg, bg = pk.geglu(proj.proj)
            g = self.linear(src, None, w=wg, bias=bg, act=nt.ACT_GEGLU)
        y3 = self.linear(g, blk.ff.net[2], residual=y2)
        self.pool.put(g, y2, box["ln"])
        return y3

    def _proj_in_with_ln(self, t, proj_in, blocks, temporal):
        """proj_in (a Linear: use_linear=True) and, where the tile holds whole rows, norm1 of the first block from the same launch
        — or, with ``fold_ln`` in a temporal transformer, the row statistics its first (folded) LayerNorm needs."""
        blk0 = blocks[0] if len(blocks) else None
        C = leaf_out_channels(proj_in)
        if self.fold_ln:
            y = self.linear(t, proj_in, want_rs=temporal and blk0 is not None and self.fold_ln_wide)
            return y, None, self.last_rs
        is_linear = isinstance(proj_in, nn.Linear) or isinstance(getattr(proj_in, "linear", None), nn.Linear)  # (or LoRA-injected)
        if blk0 is not None and is_linear and self.ln_fusable(C, blk0.norm1):
            ln1 = self.buf(t.shape[0], C)
            return self.linear(t, proj_in, ln=(blk0.norm1, ln1)), ln1, None
        return self.linear(t, proj_in), None, None

    # The transformers' GroupNorm inside proj_in's panel fill (t2v_gemm_desc::gn_coef): the statistics launch writes the per-channel
    # affine (t2v_gn_coef_cs), t2v_linear_pr applies it to the rows on their way into LDS — no apply pass, no normalised tensor.  Taken
    # at the 320-channel level, where t2v_linear_pr ties with the tuned t2v_gemm tile on the N = C launch (T2V_GN_IN=0: off).
    gn_in_fill = os.environ.get("T2V_GN_IN", "1") == "1"
    gn_in_widths = (320,)

    def _proj_in_gn_in(self, tr, x, units, rows_per_unit):
        """proj_in(GroupNorm(x)) as statistics launch + ONE t2v_linear_pr launch, or None where that form does not apply."""
        ops, norm, mod = self.ops, tr.norm, tr.proj_in
        if (not self.gn_in_fill or not self.fuse_gn or x.C not in self.gn_in_widths or len(x.parts) != 1 or x.cs[0] is None
                or not hasattr(ops, "gn_coef_cs") or not (isinstance(mod, nn.Linear) or isinstance(getattr(mod, "linear", None), nn.Linear))):
            return None
        G = norm.num_groups
        if not ops.gn_coef_cs_supported(x.cs[0], None, x.C, 0, units, rows_per_unit, G):
            return None
        w, bias = self.pk.mat(mod), self.pk.bias(mod)
        N = w.shape[0]
        coef = self.buf(units, 2 * x.C, torch.float32)
        out = self.buf(x.M, N)
        kw = dict(M=x.M, N=N, bias=bias, residual=None, act=nt.ACT_NONE, gn_in=(coef, rows_per_unit))
        if (not self.linear_pr or w.shape[1] != x.C or not hasattr(ops, "linear_pr_supported")
                or ops.linear_pr_supported(x.t, w, out, **kw) != 1):
            self.pool.put(coef, out)
            return None
        ops.gn_coef_cs(x.cs[0], None, x.C, 0, units, rows_per_unit, norm.eps, self.pk.f32(norm.weight), self.pk.f32(norm.bias), coef, G)
        ops.linear_pr(x.t, self.pk.lpr(w), out, **kw)
        self.pool.put(coef)
        return out

    def _transformer(self, tr, x, units, rows_per_unit, temporal):
        y = self._proj_in_gn_in(tr, x, units, rows_per_unit) if not self.fuse_ln else None
        if y is not None:
            ln1 = rs = None
        else:
            t = self.gn(x, tr.norm, units, rows_per_unit, False, then=self.pk.mat(tr.proj_in))
            y, ln1, rs = self._proj_in_with_ln(t, tr.proj_in, tr.transformer_blocks, temporal)
            self.pool.put(t)
        for i, blk in enumerate(tr.transformer_blocks):
            if i > 0 and self.fold_ln:
                rs = None   # (depth > 1: the previous block's feed-forward output carries no statistics; its norm1 is a launch)
            ny = self.transformer_block(blk, y, (x.n_img, x.h * x.w), temporal=temporal, ln1=ln1, rs=rs)
            ln1 = None
            self.pool.put(y)
            y = ny
        out = self.linear(y, tr.proj_out, residual=x.t, want_cs=True)   # the next layer starts with a GroupNorm
        cs = self.last_cs
        self.pool.put(y)
        return Act(out, x.n_img, x.h, x.w, cs=[cs])

    def spatial_transformer(self, st, x):
        return self._transformer(st, x, self.B * self.F, x.h * x.w, temporal=False)

    def temporal_transformer(self, tt, x):
        return self._transformer(tt, x, self.B, self.F * x.h * x.w, temporal=True)
