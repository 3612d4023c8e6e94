"""Data gradient of the UNet forward: d(loss)/d(latents) for a loss on the denoiser output and / or on the recorded
temporal attention probabilities — what ``get_motion_prior_score`` asks of autograd (``motion_prior_sample.py:59-84``:
``latents.requires_grad_(True)`` -> UNet -> ``attn1.attention_probs`` of the output-block temporal transformers ->
``autograd.grad(loss, latents)``), and the native student backward of the distillation step.  With frozen
weights only data gradients are computed, and the conditioning branch (time / fps / guidance embeddings, text K / V) carries
none because it does not depend on the latents.  With the LoRA tensors bound (``bind_lora``, engine_lora.py) the same tape also
runs the un-merged LoRA branch of every injected leaf, takes its weight gradients at the leaf's backward, applies the
train-mode dropouts, and hands d(loss)/d(emb_all) back for the conditioning branch that stays in torch.

Same construction as the VAE decoder's gradient engine: the forward is recorded together with a *tape* of closures, each
recording its block's backward launches; forward and backward are two replayable launch lists over one buffer pool.
Saved per block: GroupNorm inputs + (mean, rstd), LayerNorm inputs, q / k / v (or q, k, V^T) of every attention, the GEGLU
pre-activation.  Recomputed in the backward: attention probabilities (GEMM-formulated for the spatial layers, in-register
for the 16-frame temporal ones), everything element-wise.

  conv3x3 / (3,1,1) / 1x1 dX   the implicit-GEMM kernel on re-packed weights (flipped taps, swapped channels)
  stride-2 conv dX             zero-interleave the gradient to the input grid (``scatter2x``), then the flipped 3x3
  nearest-x2 + conv dX         flipped 3x3 at the high resolution, then a 2x2 sum-pool
  GroupNorm(+SiLU)             ``gn_bwd`` (two-part input for the skip concats; the residual branch rides in ``resid``)
  LayerNorm                    ``layernorm_bwd`` (statistics recomputed per row; residual fused)
  GEGLU                        ``geglu_bwd`` on the saved pre-activation (packed [32 value | 32 gate] column groups)
  temporal attention           ``attn_temporal_bwd`` (one wave-sized problem per pixel and head; takes d(probs) as well)
  spatial self / text cross    batched GEMMs around ``softmax_rows`` / ``softmax_bwd_rows`` with bf16 transposes; spatial self with
                               ``flash_attn_bwd``: ``attn_spatial_bwd`` (csrc/attention_bwd.hip, probabilities never in memory)

Status: the dataflow is verified on CPU against torch autograd (tests/test_unet_grad_cpu.py, emulated op backend), and the
device kernels (csrc/backward_unet.hip, train.hip, attention_bwd.hip, wgrad_tn.hip) against the emulated backend and the whole engine
against autograd on MI355X (tests/test_gpu_unet_grad.py; the same sources also run on a host SIMT simulator, tests/test_hostsim_*.py)."""
import os

import torch
import torch.nn as nn

from . import native as nt
from .native import on_tensor_device
from .engine import Act, Packer, UNetEngine, effective_weight_bias, is_lora_leaf, leaf_out_channels
from .engine_full import FullTrainMixin
from .engine_lora import LoraTrainMixin, _pad
from .unet3d import Downsample, ResBlock, SpatialTransformer, TemporalTransformer, TimestepEmbedSequential, Upsample


class UNetGradEngine(FullTrainMixin, LoraTrainMixin, UNetEngine):
    # spatial self-attention backward by the flash-style kernels of csrc/attention_bwd.hip (probabilities never in memory);
    # T2V_FLASH_ATTN_BWD=0 selects the GEMM-formulated one (both validated on MI355X; 342 -> 320 ms per distillation step)
    flash_attn_bwd = os.environ.get("T2V_FLASH_ATTN_BWD", "1") == "1"
    # activation checkpointing of the reference (lvdm/common.py:96-112 around ResBlock._forward, openaimodel3d.py:223-254, and
    # BasicTransformerBlock._forward, attention.py:300-311; yaml ``use_checkpoint: true``): with ``checkpoint_blocks`` every
    # residual block and every spatial / temporal transformer keeps only its INPUT in the forward and re-runs its forward inside
    # the backward (the counter-based dropout masks regenerate from (seed, site), so the recomputation is the same function).
    # Same gradients bit for bit, fewer live activations, one more block forward per block.  Default off: the tape of a
    # full-size step is 43 GB of 288.  T2V_NATIVE_CHECKPOINT=1 turns it on, =model follows the module's own ``use_checkpoint``
    # (what the reference's yaml sets), or set the attribute before the first forward.
    _ckpt_env = os.environ.get("T2V_NATIVE_CHECKPOINT", "0")
    _ckpt_set = None

    @property
    def checkpoint_blocks(self):
        if self._ckpt_set is not None:
            return self._ckpt_set
        if self._ckpt_env == "model":
            return bool(getattr(self.model, "use_checkpoint", False))
        return self._ckpt_env == "1"

    @checkpoint_blocks.setter
    def checkpoint_blocks(self, on):
        if bool(on) != self.checkpoint_blocks:
            self.plans.clear()                      # recorded launch lists are specific to the mode
        self._ckpt_set = bool(on)

    def __init__(self, model, ops):
        super().__init__(model, ops)
        # GroupNorm statistics of the forward from the producing GEMMs' epilogues (t2v_gemm colstat_out -> t2v_gn_stats_cs), as on
        # the inference engine; the tape keeps (mean, rstd) for the backward either way.  T2V_FUSE_GN_TRAIN=0: statistics pass
        # over the tensor (t2v_gn_stats).
        self.fuse_gn = os.environ.get("T2V_FUSE_GN_TRAIN", "1") == "1"
        # 3x3 convs on t2v_conv_halo where it takes them: the base data-gradient convs of the backward (flipped packs), and the
        # forward convs when the LoRA branch does not ride in the base leaf's epilogue (T2V_LORA_EPILOGUE=0).  Its residual add is
        # the fp32 row pass of tile80.h (one rounding), so the training numerics are those of t2v_gemm.  T2V_CONV_HALO_TRAIN=0: off.
        self.conv_halo = os.environ.get("T2V_CONV_HALO_TRAIN", "1") == "1"

    # ---- public: forward with tape, then backward ----------------------------------------------------------------
    def _active_dropouts(self):
        """Number of active Dropout(p > 0) modules; in LoRA training they must all be ones the engine (or torch's conditioning
        branch) applies: the LoRA branches' and the temporal conv blocks' (train-mode student, train_t2v_turbo_v1_lora.py:641)."""
        from .nn_util import walk_modules
        mods = walk_modules(self.model)  # (one cheap walk per call: order does not matter here)
        active = [mod for mod in mods if isinstance(mod, nn.Dropout) and mod.p > 0 and mod.training]
        if active and self.trains:
            from .unet3d import TemporalConvBlock
            known = {id(mod.dropout) for mod in mods if is_lora_leaf(mod)}
            for blk in mods:
                if isinstance(blk, TemporalConvBlock):
                    known.update(id(l) for st in (blk.conv1, blk.conv2, blk.conv3, blk.conv4) for l in st if isinstance(l, nn.Dropout))
            other = [mod for mod in active if id(mod) not in known]
            if other:
                raise RuntimeError(f"native training: {len(other)} active Dropout module(s) the engine does not apply")
        return len(active)

    @on_tensor_device
    def forward_tape(self, x, timesteps, context, fps=16, timestep_cond=None, motion_cond=None, emb_all=None, seed=None):
        """``emb_all`` (LoRA training only): the conditioning branch's output [B, sum of ResBlock widths] fp32, computed by
        the caller in torch (``conditioning_torch``) so that autograd owns that branch's 27 tiny leaves."""
        m = self.model
        assert x.dim() == 5 and context is not None
        assert (emb_all is not None) == self.trains, "emb_all is given exactly when LoRA tensors / the base weights are bound for training"
        # (full fine-tuning: every parameter moves every step — the packs are re-filled in place instead of dropping the plans)
        if not self.training_full:
            self._check_weights(m, self.lora_ids if self.training_lora else ())
        dropping = self._active_dropouts()
        if dropping and not self.trains:
            raise RuntimeError("native UNet gradient path: a Dropout(p>0) is in training mode; call .eval() first")
        key = ("grad", dropping, tuple(x.shape), x.dtype, tuple(context.shape), context.dtype, isinstance(fps, int),
               None if timestep_cond is None else tuple(timestep_cond.shape),
               None if motion_cond is None else tuple(motion_cond.shape), x.device, self.training_lora, self.training_full)
        plan = self.plans.get(key)
        if dropping:  # one seed per forward; the backward regenerates the same masks from it
            self._seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if seed is None else int(seed)
        if plan is None:
            if self.training_lora:
                # the LoRA operand / gradient arenas (engine_lora._lora_begin) are per engine, not per plan: a new input signature
                # replaces the recorded plan instead of leaving one behind whose launches point into freed arenas
                self.plans.clear()
            plan = self._own(self._record_grad(x, timesteps, context, fps, timestep_cond, motion_cond, emb_all))
            self._keep_plan(key, plan)
            if getattr(self.ops, "is_native", False):
                self._replay(plan, "rec")  # recording ran the backward once and recycled the saved buffers
        else:
            st = plan["static"]
            st["x"].copy_(x)
            st["ts"].copy_(timesteps)
            st["ctx"].copy_(context)
            if m.fps_cond:
                st["fps"].fill_(fps) if isinstance(fps, int) else st["fps"].copy_(fps)
            if timestep_cond is not None:
                st["tc"].copy_(timestep_cond)
            if motion_cond is not None:
                st["mc"].copy_(motion_cond)
            if emb_all is not None:
                st["emb_all"].copy_(emb_all.detach())
                if self.training_lora:
                    self.refresh_lora_packs()
                else:
                    self.plan = plan
                    self.full_refresh_packs()
            if dropping:
                st["seed"].fill_(self._seed)
            if "seed" in st:
                self.seed_t = st["seed"]
            self._replay(plan, "rec")
        plan["fwd_id"] = plan.get("fwd_id", 0) + 1
        self._last = plan
        self._publish_probs(plan)
        return plan["out"].clone()

    def backward(self, dout=None, dprobs=None, flat_grad=None, accumulate=True, grad_sync=None):
        """d(loss)/dx for the most recent ``forward_tape``.  ``dout``: gradient w.r.t. the output (or None = 0);
        ``dprobs``: {attention module: gradient w.r.t. its ``attention_probs``} for any of the recorded layers.
        LoRA training: the weight gradients land in ``flat_grad`` (fp32, ``bind_lora`` order; added to what is there
        when ``accumulate``) and ``self.d_emb_all`` holds d(loss)/d(emb_all) for the caller's torch branch.
        ``grad_sync``: the ``dist.FlatGradSync`` that owns ``flat_grad``.  In a multi-rank job the engine then all-reduces its
        gradient arena in segments WHILE the backward runs (engine_lora: "gradient exchange overlapped with the backward") and
        hands over already averaged gradients; ``grad_sync.all_reduce_mean()`` afterwards only exchanges the conditioning
        branch's tensors.  T2V_ASYNC_ALLREDUCE=0 keeps the single blocking all-reduce after the backward.  Under ``use_graph``
        the backward list is captured as one hipGraph per run of launches between two markers."""
        plan = self._last
        if plan["out"].is_cuda and plan["out"].device.index != torch.cuda.current_device():
            with torch.cuda.device(plan["out"].device):
                return self.backward(dout, dprobs, flat_grad, accumulate, grad_sync)
        if plan.get("bwd_id") == plan["fwd_id"]:
            raise RuntimeError("UNet gradient: backward was already run for this forward (its saved activations are gone)")
        plan["bwd_id"] = plan["fwd_id"]
        st = plan["static"]
        st["dout"].zero_() if dout is None else st["dout"].copy_(dout)
        dprobs = dprobs or {}
        known = {id(a) for a, _ in plan["probs"]}
        for a in dprobs:
            if id(a) not in known:
                raise KeyError("dprobs given for a layer that does not record attention_probs")
        for attn, _ in plan["probs"]:
            buf = plan["dprobs"][id(attn)]
            g = dprobs.get(attn)
            buf.zero_() if g is None else buf.copy_(g)
        world = self._overlap_world(grad_sync, flat_grad)
        self._overlap, self._handles = (grad_sync if world else None), []
        try:
            self._replay(plan, "rec_bwd")
        finally:
            self._overlap = None
        if self.trains:
            self.d_emb_all = plan["d_emb"]
        if self.training_lora:
            if flat_grad is not None:
                assert flat_grad.dtype == torch.float32 and flat_grad.numel() == self.lora_numel and flat_grad.is_contiguous()
                if world:
                    assert self._handles, "overlapped gradient exchange: the backward list carries no segment markers"
                    for h in self._handles:
                        h.wait()
                    self.lora_grads_into(flat_grad, accumulate, alpha=1.0 / world)
                    grad_sync.mark_engine_reduced(self.conditioning_index())
                else:
                    self.lora_grads_into(flat_grad, accumulate)
                    if grad_sync is not None and hasattr(grad_sync, "note_local_grads"):
                        grad_sync.note_local_grads()   # un-averaged: the next all_reduce_mean covers the whole buffer
        return plan["dx"].clone()

    def _overlap_world(self, grad_sync, flat_grad):
        """World size when this backward should exchange its gradient arena itself, else 0."""
        if grad_sync is None or flat_grad is None or not self.training_lora or os.environ.get("T2V_ASYNC_ALLREDUCE", "1") == "0":
            return 0
        import torch.distributed as dist
        if not dist.is_initialized() or grad_sync.flat.data_ptr() != flat_grad.data_ptr():
            return 0
        world = dist.get_world_size()
        if world == 1 and not getattr(grad_sync, "force", False):
            return 0
        return world

    def _replay(self, plan, which):
        """Replay one of the two recorded launch lists; with ``use_graph`` (T2V_HIP_GRAPH=1) each list is captured into its own
        hipGraph after its first plain replay (thousands of launches per list: the Python / ctypes loop would otherwise set
        the pace).  Everything that changes between steps lives in static device buffers (inputs, LoRA operand packs, seed)."""
        ops = self.ops
        if not getattr(ops, "is_native", False):
            plan["fn" if which == "rec" else "fn_bwd"]()
            return
        gkey, rkey = "graph_" + which, "runs_" + which
        if self.use_graph and plan.get(gkey) is not None:
            for g, host in plan[gkey]:      # graphs of the launch runs, host calls (all-reduce markers) between them
                if g is not None:
                    g.replay()
                else:
                    host[0](*host[1], None)
            return
        if self.use_graph and plan.get(rkey, 0) >= 1 and not plan.get("graph_failed") and plan.get(gkey) is None:
            try:
                # a list is cut at its host-side entries (the gradient exchange's segment markers): one hipGraph per run of
                # launches, the host calls re-issued between the graph launches
                runs, cur = [], []
                for e in plan[which]:
                    if e[2] == "allreduce_segment":
                        if cur:
                            runs.append((cur, None))
                            cur = []
                        runs.append((None, (e[0], e[1])))
                    else:
                        cur.append(e)
                if cur:
                    runs.append((cur, None))
                built = []
                torch.cuda.synchronize()
                for launches, host in runs:
                    if launches is None:
                        built.append((None, host))
                        continue
                    g = torch.cuda.CUDAGraph()
                    # (thread-local capture mode: the watchdog thread of an RCCL process group may poll its events while this thread
                    # captures — in the default global mode that invalidates the capture and aborts the watchdog)
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        ops.replay(launches, ops.stream(), cache=False)   # (a temporary sub-list: nothing to reuse)
                    built.append((g, None))
                plan[gkey] = built
                return self._replay(plan, which)
            except Exception as e:  # capture unsupported -> stay on plain replay, loudly
                plan["graph_failed"] = str(e)
                import warnings
                warnings.warn(f"hipGraph capture of the {which} list failed, replaying launches instead: {e}")
        ops.replay(plan[which], ops.stream())
        plan[rkey] = plan.get(rkey, 0) + 1

    # ---- recording ------------------------------------------------------------------------------------------------
    def _record_grad(self, x, timesteps, context, fps, timestep_cond, motion_cond, emb_all=None):
        m, ops = self.model, self.ops
        native = getattr(ops, "is_native", False)
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
        out = torch.empty(B, m.out_channels, F, H, W, dtype=x.dtype, device=x.device)
        st["dout"] = torch.zeros_like(out)
        plan = {"static": st, "out": out, "dx": torch.empty_like(st["x"]), "probs": [], "dprobs": {}, "runs": 0}
        if self.training_lora:
            # frozen base weights are packed as they are; the LoRA branch runs as its own GEMMs on per-step operand packs
            self.pk = Packer(self.adt, x.device, merge_lora=False)
            self._lora_begin()
            st["emb_all"] = emb_all.detach().to(x.device, torch.float32).clone().contiguous()
            plan["d_emb"] = torch.zeros_like(st["emb_all"])
            st["seed"] = torch.full((1,), getattr(self, "_seed", 0), dtype=torch.int64, device=x.device)
            self.seed_t = st["seed"]
        if self.training_full:
            if self.checkpoint_blocks:
                raise NotImplementedError("native full fine-tuning with checkpoint_blocks (the leaf inputs a recomputed block saves again "
                                          "are not covered by tests): leave T2V_NATIVE_CHECKPOINT off — the tape of a full-size step fits 288 GB")
            # plain leaves on packs of the current weights (re-filled in place per step: Packer.refresh); one fp32 gradient per parameter
            self.pk = Packer(self.adt, x.device)
            st["emb_all"] = emb_all.detach().to(x.device, torch.float32).clone().contiguous()
            plan["d_emb"] = torch.zeros_like(st["emb_all"])
            plan["fgrads"] = {}
            st["seed"] = torch.full((1,), getattr(self, "_seed", 0), dtype=torch.int64, device=x.device)
            self.seed_t = st["seed"]
            self._full_fp = None
            self.full_refresh_packs()
        self.plan = plan
        self.tape, self.refs = [], {}
        self._fsaved = {}

        def fwd():
            self.tape.clear()
            self.refs.clear()
            self._fsaved.clear()
            self.drop_sites = []
            plan["probs"].clear()
            self._forward_tape(st, out)

        def bwd():
            self._backward_tape(st["dout"], plan["dx"])

        if native:
            ops.init()
            for which, fn in (("rec", fwd), ("rec_bwd", bwd)):
                ops.recording = []
                try:
                    fn()
                finally:
                    plan[which] = ops.recording
                    ops.recording = None
        else:  # emulation backend (tests): the closures themselves are the plan; every forward is followed by one backward
            fwd()
            plan["fn"], plan["fn_bwd"] = fwd, bwd
        plan["pool_bytes"] = self.pool.bytes
        return plan

    # ---- saved-activation bookkeeping: a tensor may be kept by several backward closures (skip connections) ----------
    def hold(self, *ts):
        for t in ts:
            if t is not None:
                self.refs[t.data_ptr()] = self.refs.get(t.data_ptr(), 0) + 1

    def drop(self, *ts):
        for t in ts:
            if t is None:
                continue
            k = t.data_ptr()
            n = self.refs.get(k, 0) - 1
            if n > 0:
                self.refs[k] = n
            else:
                self.refs.pop(k, None)
                self.pool.put(t)

    # ---- helpers ------------------------------------------------------------------------------------------------------
    def gn_t(self, x, norm, units, rows, silu):
        """GroupNorm(+SiLU) of an Act keeping (mean, rstd); the caller keeps x for the backward."""
        ops = self.ops
        G = norm.num_groups
        stats = self.buf(units, G * 2, torch.float32)
        if self.fuse_gn and all(c is not None for c in x.cs) and rows % 32 == 0:
            ws = self.buf(1, max(ops.group_norm_cs_ws_floats(units, rows, G), 1), torch.float32)
            ops.gn_stats_cs(x.cs[0], x.cs[1] if len(x.cs) > 1 else None, x.parts[0].shape[1],
                            x.parts[1].shape[1] if len(x.parts) > 1 else 0, units, rows, norm.eps, ws, stats, G)
        else:
            ws = self.buf(1, max(ops.gn_ws_floats(units, rows, G), 1), torch.float32)
            ops.gn_stats(x.parts[0], x.p1, units, rows, norm.eps, ws, stats, G)
        out = self.buf(x.M, x.C)
        ops.gn_apply(x.parts[0], x.p1, units, rows, stats, self.pk.f32(norm.weight), self.pk.f32(norm.bias), silu, out, G)
        self.pool.put(ws)
        return out, stats

    def gn_b(self, x, norm, units, rows, silu, stats, dy, resid=None):
        """-> dx [M, C] (one buffer even when x is a virtual concat: the consumer takes column slices)."""
        ops = self.ops
        G = norm.num_groups
        if self.training_full:
            self.full_norm_grads(norm, x, dy, kind=0, silu=silu, rows_per_unit=rows, stats=stats)
        ws = self.buf(1, max(ops.gn_bwd_ws_floats(units, rows, G), 1), torch.float32)
        dx = self.buf(x.M, x.C)
        ops.gn_bwd(x.parts[0], units, rows, stats, self.pk.f32(norm.weight), self.pk.f32(norm.bias), silu, dy, resid, ws, dx, G,
                   x1=x.p1)
        self.pool.put(ws)
        return dx

    def tconv_dgrad_w(self, mod):
        """(3,1,1) conv data gradient as the same temporal conv over dy: w'[ci][(kt', co)] = w[co][ci][2 - kt']."""
        def make(out=None, ops=None):
            w = self.pk.wb(mod)[0]                                  # [co, ci, 3, 1, 1]
            if self.pk._repacked(w, out, ops, 1):
                return out
            wd = w[:, :, :, 0, 0].flip(2).permute(1, 2, 0)          # [ci, kt', co]
            return self.pk._permuted_into(wd, out)
        make.into = True
        return self.pk._memo(("tconv_dgrad", id(mod)), make)

    def mats_t(self, mods, tag):
        """Transposed pack of row-concatenated Linear weights: [K, sum N] (dx = d[y0|y1|..] @ cat(W))."""
        return self.pk._memo((tag,) + tuple(id(mm) for mm in mods),
                             lambda: self.pk.cat_mats(mods, tag + "_fwd").t().contiguous())

    def drop_site(self, drops, kind, meta=None):
        """(p, site id) of an active dropout (None if inactive).  ``drops``: the nn.Dropout module(s) this site stands for
        (one per leaf of a LoRA group); kind / meta describe the row order for tests that replay the masks in torch."""
        ps = {(d.p if d.training else 0.0) for d in drops}
        assert len(ps) == 1, "leaves of one LoRA group must share the dropout probability"
        p = ps.pop()
        if p <= 0:
            return None
        assert self.trains, "dropout is only applied on the training paths"
        self.drop_sites.append((drops, kind, meta))
        return float(p), len(self.drop_sites) - 1

    def rel(self, *ts):
        """Forward-side release of an intermediate: kept alive while a backward closure / LoRA group still holds it."""
        for t in ts:
            if t is not None and self.refs.get(t.data_ptr(), 0) == 0:
                self.pool.put(t)

    # ---- leaves: forward with the LoRA branch (training), backward with the LoRA weight gradients -----------------------
    def linear(self, a, mod, *, residual=None, act=nt.ACT_NONE, out_dtype=None, w=None, bias="auto", N=None, lora=None, perm=None,
               conv_geom=None, want_cs=False):
        """``a``: tensor or two-part Act.  ``lora``: the injected leaves whose row-concatenated weights ``w`` holds (default:
        [mod]); ``perm``: packed output row j = original row perm[j] (GEGLU packing)."""
        x = a if isinstance(a, Act) else Act(a, 0, 0, 0)
        mods = lora if lora is not None else ([mod] if mod is not None else None)
        zf = grp = None
        if self.training_full and mods:
            self.full_save(mods, x, perm=perm)
        if self.training_lora and mods and all(is_lora_leaf(mm) for mm in mods):
            assert act == nt.ACT_NONE
            meta = (self.B, self.F, x.M // (self.B * self.F)) if self.row_kind == "temporal" else (self.B, self.F, self.ctx_len)
            if conv_geom is not None:  # a 1x1 conv run as a linear layer: torch sees (n, C, h, w)
                meta = conv_geom
            grp = self.lgroup(mods, nt.GEMM_LINEAR, perm)
            self.lora_t(grp, x, x.M, kind_meta=meta, kind="conv" if conv_geom is not None else None)
        w = self.pk.mat(mod) if w is None else w
        bias = self.pk.bias(mod) if isinstance(bias, str) else bias
        N = w.shape[0] if N is None else N
        out = self.buf(x.M, N // 2 if act == nt.ACT_GEGLU else N, out_dtype)
        kw = dict(M=x.M, N=N, a1=x.p1, bias=bias, residual=residual, act=act)
        if grp is not None:
            # the LoRA branch: inside this launch's epilogue where the launch can carry it, else as up-projection launches whose
            # sum z becomes this launch's residual operand
            extra = self.lora_epilogue_args(grp) if N == grp.ntot else None
            if (extra is not None and self.lora_split_aware and hasattr(self.ops, "gemm_plan")
                    and self.ops.gemm_plan(x.parts[0], w, out, **kw)[1] > 1):
                extra = None   # (the plain launch splits K: the epilogue form, one split, would lose more than it saves)
            if extra is not None and self.ops.gemm_fuse_supported(x.parts[0], w, out, **kw, **extra):
                kw.update(extra)
            else:
                zf, kw["residual"] = self.lora_up(grp, x.M, residual)
        self.last_cs = self._colstat_for(x.parts[0], w, out, **kw) if want_cs else None   # (for the GroupNorm of the next block)
        if self.last_cs is not None:
            kw["colstat"] = self.last_cs
        self.ops.gemm(x.parts[0], w, out, **kw)
        if zf is not None:
            self.pool.put(zf)
        return out

    def conv(self, x, mod, mode, *, frames=0, rowvec=None, rowvec_div=0, residual=None, out_dtype=None, w=None, bias="auto", frozen_pack=False):
        zf = extra = None
        if w is None and self.training_full and mod is not None:
            self.full_save([mod], x, frames=frames)
        if w is None and self.training_lora and is_lora_leaf(mod):
            if mode == nt.GEMM_CONV3X3_S2:
                ho, wo = (x.h - 1) // 2 + 1, (x.w - 1) // 2 + 1
            elif mode == nt.GEMM_CONV3X3_UP2:
                ho, wo = 2 * x.h, 2 * x.w
            else:
                ho, wo = x.h, x.w
            grp = self.lgroup([mod], mode)
            self.lora_t(grp, x, x.n_img * ho * wo, frames, kind_meta=(x.n_img, ho, wo))
            extra = self.lora_epilogue_args(grp) if out_dtype in (None, self.adt) else None
            if extra is None:
                zf, residual = self.lora_up(grp, x.n_img * ho * wo, residual)
        y = super().conv(x, mod, mode, frames=frames, rowvec=rowvec, rowvec_div=rowvec_div, residual=residual,
                         out_dtype=out_dtype, w=w, bias=bias, want_cs=w is None,   # (data-gradient convs pass their own pack: no statistics)
                         frozen_pack=frozen_pack, extra=extra, fallback=None if extra is None else (lambda: self.lora_up(grp, x.n_img * ho * wo, residual)))
        if zf is not None:
            self.pool.put(zf)
        return y

    def lin_b(self, dy, w_t, residual=None, lora=None, need_dx=True, colsum=None):
        """dx = dy @ W for a [K, N]-transposed pack w_t (rows = input features); ``lora``: the leaves behind w_t — in
        training their weight gradients are taken here and their branch's data gradient joins dx."""
        if self.training_full and lora:
            self.full_linear_grads(lora, dy)
        grp = self.saved_group(lora, nt.GEMM_LINEAR)
        dl = None
        if grp is not None:
            g = self.lora_wgrad(grp, dy, colsum)
            if need_dx:
                dl = self.buf(dy.shape[0], grp.ce)
                self.ops.gemm(g, grp.Db, dl, M=dy.shape[0], N=grp.ce, residual=residual)
                residual = dl
            self.lora_wgrad_down(grp, g)
            self.pool.put(g)
        if not need_dx:
            return None
        out = self.buf(dy.shape[0], w_t.shape[0])
        self.ops.gemm(dy, w_t, out, M=dy.shape[0], N=w_t.shape[0], residual=residual)
        if dl is not None:
            self.pool.put(dl)
        return out

    def conv_b(self, dy, mod, mode, in_geom, frames=0, colsum=None, out_dtype=None, dy_pad=None):
        """Data gradient of a conv leaf: dy is an Act on the leaf's OUTPUT grid, the result an Act on its input grid
        ``in_geom`` = (n_img, h, w).  The adjoint of every gather mode is the 3x3 / (3,1,1) kernel itself on re-packed
        weights (stride 2: zero-interleaved dy; nearest-x2: 2x2 sum-pool of the result).  In training the leaf's LoRA weight
        gradients are taken here as well.  ``dy_pad``: dy with its columns zero-padded to 64 (narrow exit conv)."""
        ops = self.ops
        n, h, w = in_geom
        m_in = n * h * w
        base = self.tconv_dgrad_w(mod) if mode == nt.GEMM_TCONV3 else self.pk.conv_dgrad(mod)
        inner_mode = nt.GEMM_TCONV3 if mode == nt.GEMM_TCONV3 else nt.GEMM_CONV3X3

        def run(src, wpack, residual=None, odt=None, pool=True, frozen=False):
            """adjoint gather of ``src`` (a tensor on the output grid) against an [N, taps*c] pack (``frozen``: of frozen weights)"""
            if mode == nt.GEMM_CONV3X3_S2:
                z = self.buf(m_in, src.shape[1])
                ops.scatter2x(src, n, dy.h, dy.w, h, w, z)
                y = self.conv(Act(z, n, h, w), None, inner_mode, w=wpack, bias=None, residual=residual, out_dtype=odt)
                self.pool.put(z)
                return y.t
            if mode == nt.GEMM_CONV3X3_UP2:
                hi = self.conv(Act(src, n, 2 * h, 2 * w), None, inner_mode, w=wpack, bias=None, residual=residual)
                if not pool:
                    return hi.t
                lo = self.buf(m_in, wpack.shape[0])
                ops.sumpool2x2(hi.t, n, h, w, lo)
                self.pool.put(hi.t)
                return lo
            return self.conv(Act(src, n, h, w), None, inner_mode, frames=frames, w=wpack, bias=None, residual=residual,
                             out_dtype=odt, frozen_pack=frozen).t

        if self.training_full and dy_pad is None:
            self.full_conv_grads(mod, mode, dy.t)
        grp = self.saved_group([mod], mode)
        dl = None
        if grp is not None:
            g = self.lora_wgrad(grp, dy.t if dy_pad is None else dy_pad, colsum)
            if mode == nt.GEMM_CONV3X3_UP2:
                dl = run(g, grp.Db, pool=False)                  # LoRA branch's data gradient, high-res: pooled with the base
                G = run(g, self.sel_pack(grp.taps, grp.rp))      # rank-r gradient gathered to the input grid, per tap
                self.pool.put(g)
                self.lora_wgrad_down(grp, G)
                self.pool.put(G)
            else:  # both from one launch: columns [0, ce) = data gradient, [ce, ce + taps*rp) = gathered rank-r gradient
                dl = run(g, grp.DbSel)
                self.pool.put(g)
                self.lora_wgrad_down(grp, dl[:, grp.ce:])
        cin = base.shape[0]
        res = None if dl is None else dl[:, :cin]
        dx = run(dy.t, base, residual=res, odt=out_dtype, frozen=True)
        if dl is not None:
            self.pool.put(dl)
        return Act(dx, n, h, w)

    def add(self, a, b):
        out = self.buf(a.shape[0], a.shape[1])
        self.ops.add(a, b, out)
        return out

    # ---- forward with tape ------------------------------------------------------------------------------------------
    def _forward_tape(self, st, out):
        m, ops, pk = self.model, self.ops, self.pk
        B, F = self.B, self.F
        x = st["x"]
        _, Cin, _, H, W = x.shape
        self._conditioning(st)
        xt = self.buf(B * F * H * W, Cin)
        ops.ncfhw_to_tokens(x, xt)
        conv_in = m.input_blocks[0][0]
        c_first = leaf_out_channels(conv_in)
        h0 = self.buf(B * F * H * W, c_first)
        ops.conv_small(xt, B * F, H, W, pk.small_conv(conv_in), pk.bias(conv_in), h0)
        self.pool.put(xt)
        if self.training_full:   # the entry conv's weight gradient contracts against the latent rows, zero-padded to one 16-byte chunk
            x8 = self.buf(B * F * H * W, 8)
            ops.fill_zero(x8)
            ops.ncfhw_to_tokens(x, x8)
            self.full_save([conv_in], Act(x8, B * F, H, W), frames=0)
        if self.training_lora and is_lora_leaf(conv_in):
            # LoRA branch of the entry conv: the 4-channel latent zero-padded to one 64-channel K slab of the implicit GEMM
            x64 = self.buf(B * F * H * W, 64)
            ops.fill_zero(x64)
            ops.ncfhw_to_tokens(x, x64)
            zf, z = self.lora_z(self.lgroup([conv_in], nt.GEMM_CONV3X3), Act(x64, B * F, H, W), B * F * H * W, kind_meta=(B * F, H, W))
            h0b = self.add(h0, z)
            self.pool.put(h0, zf)
            h0 = h0b
        h = Act(h0, B * F, H, W)
        hs = []
        for i, block in enumerate(m.input_blocks):
            if i > 0:
                h = self.run_sequential_t(block, h)
            if i == 0 and m.addition_attention:
                h = self.run_sequential_t(m.init_attn, h)
            hs.append(h)
            # h is a skip connection too: in the backward, the gradient the matching output block left for it is added
            # here, i.e. after everything downstream of h on the main chain has been walked back
            self.tape.append(("join", None))
        h = self.run_sequential_t(m.middle_block, h)
        for block in m.output_blocks:
            skip = hs.pop()
            # (reverse order!) once the block below is walked back, its input gradient [M, c(h) + c(skip)] splits in two
            self.tape.append(("split", h.C))
            h = self.run_sequential_t(block, Act([h.t, skip.t], h.n_img, h.h, h.w, cs=[h.cs[0], skip.cs[0]]))
        xin = h
        self.hold(xin.t)
        tt, st_out = self.gn_t(xin, m.out[0], B * F, H * W, True)
        y = self.conv(Act(tt, h.n_img, h.h, h.w), m.out[2], nt.GEMM_CONV3X3, out_dtype=torch.float32)
        self.rel(tt)
        ops.tokens_to_ncfhw(y.t, out)
        self.pool.put(y.t)
        n_img = B * F

        cpad = 4 if m.out_channels <= 4 else 8
        assert m.out_channels <= 8, "the direct small-channel conv handles up to 8 gradient channels"

        def exit_bwd(dout):
            # conv_out data gradient (4 -> 320 channels) on the direct small-channel conv, like the VAE decoder's exit
            d4 = self.buf(n_img * H * W, cpad)
            if cpad != m.out_channels:
                ops.fill_zero(d4)
            ops.ncfhw_to_tokens(dout, d4)
            dt = self.buf(n_img * H * W, xin.C)
            ops.conv_small(d4, n_img, H, W, pk.small_conv_dgrad(m.out[2], cin_pad=cpad), None, dt)
            self.pool.put(d4)
            if self.training_full:   # the exit conv's own weight / bias gradient: dy zero-padded to one 16-byte chunk of columns
                d8 = self.buf(n_img * H * W, 8)
                ops.fill_zero(d8)
                ops.ncfhw_to_tokens(dout, d8)
                self.full_conv_grads(m.out[2], nt.GEMM_CONV3X3, d8)
                self.pool.put(d8)
            grp = self.saved_group([m.out[2]], nt.GEMM_CONV3X3)
            if grp is not None:  # LoRA of the exit conv: dy zero-padded to 64 columns (K of  g = s dy U)
                d64 = self.buf(n_img * H * W, 64)
                ops.fill_zero(d64)
                ops.ncfhw_to_tokens(dout, d64)
                g = self.lora_wgrad(grp, d64)
                self.pool.put(d64)
                dl = self.conv(Act(g, n_img, H, W), None, nt.GEMM_CONV3X3, w=grp.DbSel, bias=None).t
                self.pool.put(g)
                self.lora_wgrad_down(grp, dl[:, grp.ce:])
                dt2 = self.add(dt, dl[:, :grp.ce])
                self.pool.put(dt, dl)
                dt = dt2
            dx = self.gn_b(xin, m.out[0], n_img, H * W, True, st_out, dt)
            self.pool.put(dt, st_out)
            self.drop(xin.t)
            return Act(dx, n_img, H, W)

        def entry_bwd(dy, dx_out):
            # conv_in data gradient (320 -> 4 channels): a narrow-N implicit GEMM like the forward's conv_out, fp32 out
            d = self.conv_b(dy, conv_in, nt.GEMM_CONV3X3, (n_img, H, W), out_dtype=torch.float32)
            self._free_view(dy.t)
            ops.tokens_to_ncfhw(d.t, dx_out)
            self.pool.put(d.t)

        self.exit_bwd, self.entry_bwd = exit_bwd, entry_bwd

    def _conditioning(self, st):
        """Embedding / context branch of UNetEngine._forward (no dependence on the latents: no tape)."""
        m, ops, pk = self.model, self.ops, self.pk
        B = self.B
        mc = m.model_channels
        L, D = st["ctx"].shape[1], st["ctx"].shape[2]
        resblocks = [mod for mod in m.modules() if isinstance(mod, ResBlock)]
        self.emb_off, off = {}, 0
        for rb in resblocks:
            self.emb_off[id(rb)] = off
            off += rb.out_channels
        self.ctx = self.buf(B * L, D)
        ops.cast(st["ctx"], self.ctx)
        self.ctx_len = L
        self.ctx_kv = {}
        self._ctx_f = None
        if self.trains:  # the M = B-row branch belongs to torch autograd (engine_lora.py / engine_full.py); its result is an input
            assert st["emb_all"].shape == (B, off)
            self.emb_all = st["emb_all"]
            return
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
        lins = [rb.emb_layers[1] for rb in resblocks]
        w_all = pk.cat_mats(lins, "emb_all")
        b_all = pk._memo(("emb_all_bias",) + tuple(id(l) for l in lins),
                         lambda: torch.cat([pk.bias(l) for l in lins]).contiguous())
        self.emb_all = self.linear(emb_s, None, w=w_all, bias=b_all, out_dtype=torch.float32)

    def context_kv_t(self, attn, per_frame=False):
        """Training: K / V of ONE cross-attention layer through its own injected to_k / to_v (token-major rows of the text
        context), V^T by a per-image transpose.  (Inference stacks all 16 layers' projections into two GEMMs.)
        ``per_frame``: project the context repeated per frame, as the reference does — needed when the LoRA branches' dropout
        is active, so that every frame draws its own masks."""
        ops = self.ops
        B, L = self.B, self.ctx_len
        inner = attn.heads * attn.dim_head
        kp = _pad(L, 64)
        src, n_sets = self.ctx, B
        if per_frame:
            src, n_sets = self.context_per_frame(), B * self.F
        kind, self.row_kind = self.row_kind, ("rows" if per_frame else "ctx")
        k = self.linear(src, attn.to_k, bias=None)
        v = self.linear(src, attn.to_v, bias=None)
        self.row_kind = kind
        vt = self.buf(n_sets * inner, kp)
        ops.fill_zero(vt)
        ops.transpose(v, L, inner, vt, batch=n_sets, in_stride=L * inner, out_stride=inner * kp)
        self.pool.put(v)
        return k, vt, kp, inner * kp

    def context_per_frame(self):
        """The text context with each clip's rows repeated for its F frames ([(b f) l, D]; context.repeat_interleave of
        openaimodel3d.py:710), built once per forward."""
        if getattr(self, "_ctx_f", None) is None:
            B, F, L = self.B, self.F, self.ctx_len
            self._ctx_f = self.buf(B * F * L, self.ctx.shape[1])
            for b in range(B):
                for f in range(F):
                    self.ops.cast(self.ctx[b * L:(b + 1) * L], self._ctx_f[(b * F + f) * L:(b * F + f + 1) * L])
        return self._ctx_f

    def _backward_tape(self, dout, dx_out):
        if self.training_lora:
            self._seg_reset()
        d = self.exit_bwd(dout)
        skip_grads = []  # gradients of the skip tensors, in the order the output blocks produced them
        for kind, fn in reversed(self.tape):
            if kind == "split":
                c0 = fn
                # d is [M, c0 + c_skip]: the first c0 columns continue down the chain, the rest wait for their producer
                skip_grads.append((d.t, c0))
                d = Act(d.t[:, :c0], d.n_img, d.h, d.w)
            elif kind == "join":
                # the tensor this block produced was also a skip connection: add that gradient before going on
                buf, c0 = skip_grads.pop()
                total = self.add(d.t, buf[:, c0:])
                self._free_view(d.t)
                self.pool.put(buf)
                d = Act(total, d.n_img, d.h, d.w)
            else:
                d = fn(d)
        self.entry_bwd(d, dx_out)
        if self.training_lora:
            self._seg_emit_down_to(0)   # whatever is left of the arena (and every piece, if some group never finished)

    def _free_view(self, t):
        """Release a gradient tensor unless it is a column slice of a concat-gradient buffer (freed with the buffer)."""
        if t.data_ptr() in self.pool.live and t.stride(0) == t.shape[1]:
            self.pool.put(t)

    def checkpointed(self, block_t, layer, h):
        """``block_t(layer, h)`` (one tape entry) with the reference's checkpoint semantics: see ``checkpoint_blocks``."""
        if not self.checkpoint_blocks:
            return block_t(layer, h)
        pool, n_probs = self.pool, len(self.plan["probs"])
        if self.training_lora and self._active_dropouts():
            self.context_per_frame()                # shared by every cross-attention: must not belong to one block
        live0, refs0 = set(pool.live), dict(self.refs)
        n_tape, n_sites = len(self.tape), len(self.drop_sites)
        saved0 = {k: g.saved for k, g in getattr(self, "_groups", {}).items()}
        y = block_t(layer, h)
        assert len(self.tape) == n_tape + 1 and self.tape[-1][0] == "block"
        if len(self.plan["probs"]) != n_probs:
            raise NotImplementedError("native checkpointing of a block that records attention_probs (record_attn_probs)")
        del self.tape[n_tape:]
        # nothing the block kept survives the forward, except its output (the next block's input) ...
        keep = {p.data_ptr() for p in y.parts}
        for q in list(keep):                        # (with the column statistics its producing GEMM wrote next to it)
            keep.update(side.data_ptr() for side in pool.links.get(q, ()))
        ctx_f = getattr(self, "_ctx_f", None)
        if ctx_f is not None:
            keep.add(ctx_f.data_ptr())
        for ptr in [q for q in pool.live if q not in live0 and q not in keep]:
            pool.release_ptr(ptr)
        # ... and what it holds of the tensors that were there before it (its input, the shared text context): those holds stay
        # until this block's backward has run, exactly as on the tape
        extra = {q: n - refs0.get(q, 0) for q, n in self.refs.items() if q in live0 and n > refs0.get(q, 0)}
        self.refs.clear()
        self.refs.update(refs0)
        for q, n in extra.items():
            self.refs[q] = self.refs.get(q, 0) + n
        for k, g in getattr(self, "_groups", {}).items():
            if g.saved is not saved0.get(k):
                g.saved = None
        sites_fwd = self.drop_sites[n_sites:]

        def bwd(dy):
            outer_tape, outer_sites = self.tape, self.drop_sites
            self.tape, self.drop_sites = [], outer_sites[:n_sites]   # the block's dropout sites get the numbers they had
            y2 = block_t(layer, h)
            assert [(k, m) for _, k, m in self.drop_sites[n_sites:]] == [(k, m) for _, k, m in sites_fwd]
            sub, self.tape, self.drop_sites = self.tape, outer_tape, outer_sites
            for p in y2.parts:                      # the recomputed output itself is not needed
                if self.refs.get(p.data_ptr(), 0) == 0 and p.data_ptr() in pool.live:
                    pool.put(p)
            d = dy
            for kind, fn in reversed(sub):
                assert kind == "block"
                d = fn(d)
            for q, n in extra.items():              # the first forward's holds (the recomputation took and dropped its own)
                left = self.refs.get(q, 0) - n
                if left > 0:
                    self.refs[q] = left
                else:
                    self.refs.pop(q, None)
                    pool.release_ptr(q)
            return d

        self.tape.append(("block", bwd))
        return y

    def run_sequential_t(self, seq, h):
        assert isinstance(seq, TimestepEmbedSequential)
        for layer in seq:
            if isinstance(layer, ResBlock):
                h = self.checkpointed(self.res_block_t, layer, h)
            elif isinstance(layer, SpatialTransformer):
                h = self.checkpointed(self.spatial_transformer_t, layer, h)
            elif isinstance(layer, TemporalTransformer):
                h = self.checkpointed(self.temporal_transformer_t, layer, h)
            elif isinstance(layer, Downsample):
                assert layer.use_conv
                h = self.downsample_t(layer, h)
            elif isinstance(layer, Upsample):
                assert layer.use_conv
                src = h
                h = self.upsample_t(layer, h)
                if self.refs.get(src.t.data_ptr(), 0) == 0:  # an intermediate nobody keeps for the backward
                    self.pool.put(src.t)
            else:
                raise NotImplementedError(f"native gradient path: unsupported layer {type(layer).__name__}")
        return h

    # ---- blocks ----------------------------------------------------------------------------------------------------------
    def res_block_t(self, rb, x):
        B, F = self.B, self.F
        n, hw = x.n_img, x.h * x.w
        off, cout = self.emb_off[id(rb)], rb.out_channels
        self.hold(*x.parts)
        t1, st1 = self.gn_t(x, rb.in_layers[0], B * F, hw, True)
        h1 = self.conv(Act(t1, n, x.h, x.w), rb.in_layers[2], nt.GEMM_CONV3X3,
                       rowvec=self.emb_all[:, off:off + cout], rowvec_div=F * hw)
        self.rel(t1)
        t2, st2 = self.gn_t(h1, rb.out_layers[0], B * F, hw, True)
        identity = isinstance(rb.skip_connection, nn.Identity)
        if identity:
            skip, own = x.t, False
        else:
            sc = rb.skip_connection
            assert effective_weight_bias(sc)[0].shape[-1] == 1, "3x3 skip convs are not built by the VideoCrafter2 config"
            skip = self.linear(x, sc, conv_geom=(n, x.h, x.w))
            own = True
        h2 = self.conv(Act(t2, n, x.h, x.w), rb.out_layers[3], nt.GEMM_CONV3X3, residual=skip)
        self.rel(t2)
        if own:
            self.pool.put(skip)
        geom = (n, x.h, x.w)
        tc_bwd = None
        y = h2
        if rb.use_temporal_conv:
            y, tc_bwd = self.temporal_conv_block_t(rb.temopral_conv, h2)

        def bwd(dy):
            d_h2 = tc_bwd(dy) if tc_bwd is not None else dy.t
            d_t2 = self.conv_b(Act(d_h2, *geom), rb.out_layers[3], nt.GEMM_CONV3X3, geom)
            d_h1 = self.gn_b(h1, rb.out_layers[0], B * F, hw, True, st2, d_t2.t)
            self.pool.put(d_t2.t, h1.t, st2)
            # training: the per-clip column sums of d_h1 are d(loss)/d(emb_all) of this block (rowvec of the forward conv)
            colsum = self.plan["d_emb"][:, off:off + cout] if self.training_lora else None
            if self.training_full:
                self.full_colsum(d_h1, self.plan["d_emb"][:, off:off + cout], F * hw)
            d_t1 = self.conv_b(Act(d_h1, *geom), rb.in_layers[2], nt.GEMM_CONV3X3, geom, colsum=colsum)
            self.pool.put(d_h1)
            d_skip = d_h2 if identity else self.lin_b(d_h2, self.pk.mat_t(rb.skip_connection), lora=[rb.skip_connection])
            dx = self.gn_b(x, rb.in_layers[0], B * F, hw, True, st1, d_t1.t, resid=d_skip)
            self.pool.put(d_t1.t, st1)
            if not identity:
                self.pool.put(d_skip)
            self._free_view(d_h2)
            self.drop(*x.parts)
            return Act(dx, *geom)

        self.tape.append(("block", bwd))
        return y

    def temporal_conv_block_t(self, tc, h2):
        """-> (output, backward closure: dy Act -> gradient w.r.t. h2 as a tensor)."""
        B, F = self.B, self.F
        n, hw = h2.n_img, h2.h * h2.w
        geom = (n, h2.h, h2.w)
        stages = (tc.conv1, tc.conv2, tc.conv3, tc.conv4)
        saved = []
        y = h2
        for i, stage in enumerate(stages):
            tt, st = self.gn_t(y, stage[0], B, F * hw, True)
            site = self.drop_site([l for l in stage if isinstance(l, nn.Dropout)], "tconv", (B, F, h2.h, h2.w)) if len(stage) > 3 else None
            if site is not None:  # train-mode student: Dropout between SiLU and the (3,1,1) conv (openaimodel3d.py:280-297)
                self.ops.dropout(tt, None, tt, tt.shape[1], site[0], self.seed_t, site[1])
            ny = self.conv(Act(tt, *geom), stage[-1], nt.GEMM_TCONV3, frames=F, residual=h2.t if i == 3 else None)
            self.rel(tt)
            saved.append((y, st, site))
            y = ny

        def bwd(dy):
            d = dy.t
            for i in (3, 2, 1, 0):
                yi, st, site = saved[i]
                d_tt = self.conv_b(Act(d, *geom), stages[i][-1], nt.GEMM_TCONV3, geom, frames=F)
                if site is not None:
                    self.ops.dropout(d_tt.t, None, d_tt.t, d_tt.t.shape[1], site[0], self.seed_t, site[1])
                if i != 3:
                    self.pool.put(d)
                nd = self.gn_b(yi, stages[i][0], B, F * hw, True, st, d_tt.t, resid=dy.t if i == 0 else None)
                self.pool.put(d_tt.t, st)
                if i != 0:
                    self.pool.put(yi.t)  # h2 itself (i == 0) belongs to the residual block's closure
                d = nd
            self._free_view(dy.t)
            self.pool.put(h2.t)
            return d

        return y, bwd

    def downsample_t(self, layer, x):
        y = self.conv(x, layer.op, nt.GEMM_CONV3X3_S2)
        n, h, w, cin = x.n_img, x.h, x.w, x.C

        def bwd(dy):
            dx = self.conv_b(dy, layer.op, nt.GEMM_CONV3X3_S2, (n, h, w))
            self._free_view(dy.t)
            return dx

        self.tape.append(("block", bwd))
        return y

    def upsample_t(self, layer, x):
        y = self.conv(x, layer.conv, nt.GEMM_CONV3X3_UP2)
        n, h, w, cin = x.n_img, x.h, x.w, x.C

        def bwd(dy):
            dx = self.conv_b(dy, layer.conv, nt.GEMM_CONV3X3_UP2, (n, h, w))
            self._free_view(dy.t)
            return dx

        self.tape.append(("block", bwd))
        return y

    def spatial_transformer_t(self, st, x):
        return self._transformer_t(st, x, temporal=False)

    def temporal_transformer_t(self, tt, x):
        return self._transformer_t(tt, x, temporal=True)

    def _transformer_t(self, tr, x, temporal):
        B, F = self.B, self.F
        n, hw = x.n_img, x.h * x.w
        units, rows = (B, F * hw) if temporal else (B * F, hw)
        self.row_kind = "temporal" if temporal else "rows"
        self.hold(x.t)
        t, stats = self.gn_t(x, tr.norm, units, rows, False)
        y = self.linear(t, tr.proj_in)
        self.rel(t)
        blocks = []
        for blk in tr.transformer_blocks:
            y, b = self.transformer_block_t(blk, y, (n, hw), temporal)
            blocks.append(b)
        out = self.linear(y, tr.proj_out, residual=x.t, want_cs=True)
        out_cs = self.last_cs
        self.rel(y)
        geom = (n, x.h, x.w)

        def bwd(dy):
            d = self.lin_b(dy.t, self.pk.mat_t(tr.proj_out), lora=[tr.proj_out])
            for b in reversed(blocks):
                d = b(d)
            d_t = self.lin_b(d, self.pk.mat_t(tr.proj_in), lora=[tr.proj_in])
            self.pool.put(d)
            dx = self.gn_b(x, tr.norm, units, rows, False, stats, d_t, resid=dy.t)
            self.pool.put(d_t, stats)
            self._free_view(dy.t)
            self.drop(x.t)
            return Act(dx, *geom)

        self.tape.append(("block", bwd))
        return Act(out, *geom, cs=[out_cs])

    def transformer_block_t(self, blk, y, x_geom, temporal):
        """-> (output rows, backward closure: d(output) tensor -> d(input) tensor).  The block's input y stays alive
        until the closure ran; the closure frees it."""
        ops, pk = self.ops, self.pk
        B, F = self.B, self.F
        M, C = y.shape
        n_img, hw = x_geom
        a1, a2 = blk.attn1, blk.attn2
        self._check_heads(a1)
        inner = a1.heads * a1.dim_head

        def lnorm(norm, src):
            ln = self.buf(M, C)
            ops.layernorm(src, pk.f32(norm.weight), pk.f32(norm.bias), norm.eps, ln)
            return ln

        def ln_b(norm, src, d_ln, resid):
            if self.training_full:
                self.full_norm_grads(norm, src, d_ln, kind=1)
            dx = self.buf(M, C)
            ops.layernorm_bwd(src, pk.f32(norm.weight), norm.eps, d_ln, resid, dx)
            return dx

        # ---- attention flavours: forward returns (o, backward: d_o -> d(ln input of the attention)) ------------------
        def temporal_attn(attn, src):
            mods = [attn.to_q, attn.to_k, attn.to_v]
            qkv = self.linear(src, None, w=pk.cat_mats(mods, "qkv"), bias=None, lora=mods)
            o = self.buf(M, inner)
            probs = dprobs = None
            if attn.record_attn_probs:
                probs = torch.empty(B * hw * attn.heads, F, F, dtype=torch.float32, device=self.device)
                self.plan["probs"].append((attn, probs))
                dprobs = self.plan["dprobs"].setdefault(id(attn), torch.zeros_like(probs))
            ops.attn_temporal(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], o, B, F, hw, attn.heads, attn.scale, probs)

            def bwd(d_o):
                dqkv = self.buf(M, 3 * inner)
                ops.attn_temporal_bwd(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], d_o, dprobs,
                                      dqkv[:, :inner], dqkv[:, inner:2 * inner], dqkv[:, 2 * inner:], B, F, hw, attn.heads, attn.scale)
                self.pool.put(qkv, d_o)
                d_ln = self.lin_b(dqkv, self.mats_t(mods, "qkv_t"), lora=mods)
                self.pool.put(dqkv)
                return d_ln
            return o, bwd

        def spatial_self_attn(attn, src):
            heads = attn.heads
            qk = self.linear(src, None, w=pk.cat_mats([attn.to_q, attn.to_k], "qk"), bias=None, lora=[attn.to_q, attn.to_k])
            kp = ((hw + 63) // 64) * 64
            vt = self.buf(n_img * inner, kp)
            if kp != hw:
                ops.fill_zero(vt)
            if (self.training_lora and is_lora_leaf(attn.to_v)) or self.training_full:
                # training: V token-major through the injected leaf, then transposed per image (inference: V^T straight
                # out of a GEMM with the weight as the row operand)
                v = self.linear(src, attn.to_v, bias=None)
                ops.transpose(v, hw, inner, vt, batch=n_img, in_stride=hw * inner, out_stride=inner * kp)
                if not self.flash_attn_bwd:
                    self.pool.put(v)
                    v = None
            else:
                v = None
                ops.gemm(pk.mat(attn.to_v), src, vt, M=inner, N=hw, batch=n_img, w_strides=(hw * src.stride(0), 0), o_strides=(inner * kp, 0))
            o = self.buf(M, inner)
            ops.attn_spatial(qk[:, :inner], qk[:, inner:], vt, kp, o, n_img, hw, hw, heads, 1, attn.scale)
            nb = n_img * heads
            if self.flash_attn_bwd:
                self.hold(o)  # the flash-style backward needs the forward's output (D = rowsum(dO * O))

            def bwd_flash(d_o):
                """dQ / dK / dV without the [queries x keys] matrices in memory (csrc/attention_bwd.hip): two launches + the
                three small operand transposes, instead of 5 batched GEMMs over 2 x 1 GB matrices, 2 softmax passes and 4 large
                transposes at the 2560-token level."""
                q, k = qk[:, :inner], qk[:, inner:]
                ld = qk.stride(0)
                if v is not None:   # token-major V of the training path
                    v_buf, vis, vhs = v, hw * inner, 64
                else:               # per (image, head) [keys][64]: the transpose of the forward's V^T rows
                    v_buf = self.buf(nb * kp, 64)
                    ops.transpose(vt, 64, kp, v_buf, batch=nb, in_stride=64 * kp, out_stride=kp * 64)
                    vis, vhs = heads * kp * 64, kp * 64
                kT = self.tposed(k, hw, inner, batch=n_img, in_stride=hw * ld)
                qT = self.tposed(q, hw, inner, batch=n_img, in_stride=hw * ld)
                doT = self.tposed(d_o, hw, inner, batch=n_img, in_stride=hw * inner)
                l2, dsum = self.buf(nb, kp, torch.float32), self.buf(nb, kp, torch.float32)
                dqk, d_v = self.buf(M, 2 * inner), self.buf(M, inner)
                ops.attn_spatial_bwd(q, k, v_buf, vis, vhs, kT, qT, doT, d_o, o, l2, dsum, dqk[:, :inner], dqk[:, inner:], d_v,
                                     n_img, hw, heads, attn.scale)
                self.pool.put(kT, qT, doT, l2, dsum, v_buf, vt, qk, d_o)
                self.drop(o)
                d1 = self.lin_b(dqk, self.mats_t([attn.to_q, attn.to_k], "qk_t"), lora=[attn.to_q, attn.to_k])
                d_ln = self.lin_b(d_v, pk.mat_t(attn.to_v), residual=d1, lora=[attn.to_v])
                self.pool.put(dqk, d_v, d1)
                return d_ln

            def tposed(src2d, rows, cols, in_stride, batch):
                """[batch][rows][cols] -> [batch][cols][kp] (rows padded with zeros up to kp)."""
                dst = self.buf(batch * cols, kp)
                if kp != rows:
                    ops.fill_zero(dst)
                ops.transpose(src2d, rows, cols, dst, batch=batch, in_stride=in_stride, out_stride=cols * kp)
                return dst

            def bwd(d_o):
                q, k = qk[:, :inner], qk[:, inner:]
                ld = qk.stride(0)
                hb = dict(batch=nb, batch_inner=heads)
                # P = softmax(scale * Q K^T) per (image, head): rows (img, head, q), kp columns
                s = self.buf(nb * hw, kp)
                if kp != hw:
                    ops.fill_zero(s)
                ops.gemm(q[:, :64], k[:, :64], s, M=hw, N=hw, alpha=attn.scale, a_strides=(hw * ld, 64), w_strides=(hw * ld, 64),
                         o_strides=(heads * hw * kp, hw * kp), **hb)
                ops.softmax_rows(s, nb * hw, hw, kp, kp)
                # dP[q][kv] = sum_c dO[q][c] V[kv][c]: W = V token-major per head = transpose of the V^T rows
                v_tok = self.buf(nb * kp, 64)
                ops.transpose(vt, 64, kp, v_tok, batch=nb, in_stride=64 * kp, out_stride=kp * 64)
                dp = self.buf(nb * hw, kp)
                ops.gemm(d_o[:, :64], v_tok, dp, M=hw, N=kp, a_strides=(hw * inner, 64), w_strides=(heads * kp * 64, kp * 64),
                         o_strides=(heads * hw * kp, hw * kp), **hb)
                self.pool.put(v_tok, vt)
                # dV[kv][c] = sum_q P[q][kv] dO[q][c]: A = P^T [kv][q], W = dO^T [c][q] (one transpose per image covers all heads)
                pT = tposed(s, hw, kp, hw * kp, nb)                       # [nb][kp][kp]
                doT = tposed(d_o, hw, inner, hw * inner, n_img)           # [n_img][inner][kp] = [nb][64][kp]
                self.pool.put(d_o)
                d_v = self.buf(M, inner)
                ops.gemm(pT, doT, d_v[:, :64], M=hw, N=64, a_strides=(heads * kp * kp, kp * kp), w_strides=(inner * kp, 64 * kp),
                         o_strides=(hw * inner, 64), **hb)
                self.pool.put(pT, doT)
                ops.softmax_bwd_rows(s, dp, nb * hw, hw, kp, kp)         # dS in place on dP
                self.pool.put(s)
                dqk = self.buf(M, 2 * inner)
                kT = tposed(k, hw, inner, hw * ld, n_img)
                ops.gemm(dp, kT, dqk[:, :64], M=hw, N=64, alpha=attn.scale, a_strides=(heads * hw * kp, hw * kp),
                         w_strides=(inner * kp, 64 * kp), o_strides=(hw * 2 * inner, 64), **hb)
                self.pool.put(kT)
                dsT = tposed(dp, hw, kp, hw * kp, nb)
                qT = tposed(q, hw, inner, hw * ld, n_img)
                self.pool.put(dp, qk)
                ops.gemm(dsT, qT, dqk[:, inner:inner + 64], M=hw, N=64, alpha=attn.scale, a_strides=(heads * kp * kp, kp * kp),
                         w_strides=(inner * kp, 64 * kp), o_strides=(hw * 2 * inner, 64), **hb)
                self.pool.put(dsT, qT)
                d1 = self.lin_b(dqk, self.mats_t([attn.to_q, attn.to_k], "qk_t"), lora=[attn.to_q, attn.to_k])
                d_ln = self.lin_b(d_v, pk.mat_t(attn.to_v), residual=d1, lora=[attn.to_v])
                self.pool.put(dqk, d_v, d1)
                return d_ln
            return o, (bwd_flash if self.flash_attn_bwd else bwd)

        def cross_attn(attn, src):
            heads, L = attn.heads, self.ctx_len
            q = self.linear(src, attn.to_q, bias=None)
            own_kv = (self.training_lora and is_lora_leaf(attn.to_k) and is_lora_leaf(attn.to_v)) or self.training_full
            # The frames of a clip share the text K / V (one attention of F*hw queries over L keys per clip and head) — unless
            # the train-mode student's dropout sits on the to_k / to_v LoRA branches: the reference projects the context
            # REPEATED per frame (openaimodel3d.py:710), so every frame draws its own masks and has its own K / V.
            per_frame = own_kv and self.training_lora and any(d.training and d.p > 0 for d in (attn.to_k.dropout, attn.to_v.dropout))
            nf = F if per_frame else 1           # K / V sets per clip
            mq = hw if per_frame else F * hw     # queries per K / V set
            if own_kv:
                k, vt, kp, vt_stride = self.context_kv_t(attn, per_frame)
            else:
                k, vt, kp, vt_stride = self.context_kv(attn)
            o = self.buf(M, inner)
            ops.attn_spatial(q, k, vt, kp, o, n_img, hw, L, heads, 1 if per_frame else F, attn.scale, vt_stride)

            def bwd(d_o):
                d_q = self.buf(M, inner)
                dk = dv = None
                if own_kv:  # the text context itself carries no gradient, its injected to_k / to_v projections do
                    dk, dv = self.buf(B * nf * L, inner), self.buf(B * nf * L, inner)
                    mqp = _pad(mq, 64)
                ldk = k.stride(0)
                hb = dict(batch=nf * heads, batch_inner=heads)  # z0 = K / V set (frame) of the clip, z1 = head
                for b in range(B):
                    rows = slice(b * F * hw, (b + 1) * F * hw)
                    qb, dob, dqb = q[rows], d_o[rows], d_q[rows]
                    kb = k[b * nf * L:(b + 1) * nf * L]
                    vtb = torch.as_strided(vt, (nf * inner, kp), (kp, 1), vt.storage_offset() + b * nf * vt_stride)
                    s = self.buf(nf * heads * mq, kp)
                    ops.fill_zero(s)
                    so = (heads * mq * kp, mq * kp)
                    ops.gemm(qb[:, :64], kb[:, :64], s, M=mq, N=L, alpha=attn.scale, a_strides=(mq * inner, 64), w_strides=(L * ldk, 64),
                             o_strides=so, **hb)
                    ops.softmax_rows(s, nf * heads * mq, L, kp, kp)
                    v_tok = self.buf(nf * heads * kp, 64)
                    ops.transpose(vtb, 64, kp, v_tok, batch=nf * heads, in_stride=64 * kp, out_stride=kp * 64)
                    dp = self.buf(nf * heads * mq, kp)
                    ops.gemm(dob[:, :64], v_tok, dp, M=mq, N=kp, a_strides=(mq * inner, 64), w_strides=(heads * kp * 64, kp * 64),
                             o_strides=so, **hb)
                    self.pool.put(v_tok)
                    kv_rows = slice(b * nf * L, (b + 1) * nf * L)
                    wg = dict(a_strides=(heads * kp * mqp, kp * mqp), w_strides=(inner * mqp, 64 * mqp), o_strides=(L * inner, 64),
                              split_k=self.split_for(L * heads, 64, mqp), **hb) if own_kv else None
                    if own_kv:  # dV[kv][c] = sum_q P[q][kv] dO[q][c], contraction over all queries of the K / V set
                        pT = self.tposed(s, mq, kp, batch=nf * heads, in_stride=mq * kp)
                        doT = self.tposed(dob, mq, inner, batch=nf, in_stride=mq * inner)
                        ops.gemm(pT, doT, dv[kv_rows][:, :64], M=L, N=64, **wg)
                        self.pool.put(pT, doT)
                    ops.softmax_bwd_rows(s, dp, nf * heads * mq, L, kp, kp)
                    self.pool.put(s)
                    kT = self.buf(nf * inner, kp)
                    ops.fill_zero(kT)
                    ops.transpose(kb, L, inner, kT, batch=nf, in_stride=L * ldk, out_stride=inner * kp)
                    ops.gemm(dp, kT, dqb[:, :64], M=mq, N=64, alpha=attn.scale, a_strides=so, w_strides=(inner * kp, 64 * kp),
                             o_strides=(mq * inner, 64), **hb)
                    self.pool.put(kT)
                    if own_kv:  # dK[kv][c] = scale * sum_q dS[q][kv] Q[q][c]
                        dsT = self.tposed(dp, mq, kp, batch=nf * heads, in_stride=mq * kp)
                        qT = self.tposed(qb, mq, inner, batch=nf, in_stride=mq * inner)
                        ops.gemm(dsT, qT, dk[kv_rows][:, :64], M=L, N=64, alpha=attn.scale, **wg)
                        self.pool.put(dsT, qT)
                    self.pool.put(dp)
                self.pool.put(q, d_o)
                if own_kv:
                    self.lin_b(dk, None, lora=[attn.to_k], need_dx=False)
                    self.lin_b(dv, None, lora=[attn.to_v], need_dx=False)
                    self.pool.put(dk, dv, k, vt)
                d_ln = self.lin_b(d_q, pk.mat_t(attn.to_q), lora=[attn.to_q])
                self.pool.put(d_q)
                return d_ln
            return o, bwd

        # ---- forward ------------------------------------------------------------------------------------------------
        ln = lnorm(blk.norm1, y)
        o, attn1_b = temporal_attn(a1, ln) if temporal else spatial_self_attn(a1, ln)
        self.rel(ln)
        y1 = self.linear(o, a1.to_out[0], residual=y)
        self.rel(o)
        ln = lnorm(blk.norm2, y1)
        o, attn2_b = temporal_attn(a2, ln) if temporal else cross_attn(a2, ln)
        self.rel(ln)
        y2 = self.linear(o, a2.to_out[0], residual=y1)
        self.rel(o)
        ln = lnorm(blk.norm3, y2)
        proj = blk.ff.net[0]
        assert hasattr(proj, "proj"), "non-gated FeedForward is not built by the VideoCrafter2 config"
        wg, bg = pk.geglu(proj.proj)
        ff_inner = wg.shape[0] // 2
        j = torch.arange(2 * ff_inner)  # packed GEGLU row j (64-row groups [32 value | 32 gate]) <- original projection row
        geglu_perm = (j // 64) * 32 + (j % 32) + (j % 64 >= 32) * ff_inner
        # pre-activation kept for the backward (packed value | gate groups)
        hpre = self.linear(ln, None, w=wg, bias=bg, lora=[proj.proj], perm=geglu_perm)
        self.rel(ln)
        g = self.buf(M, hpre.shape[1] // 2)
        ops.geglu_fwd(hpre, g)
        y3 = self.linear(g, blk.ff.net[2], residual=y2)
        self.rel(g)

        def bwd(dy3):
            d_g = self.lin_b(dy3, pk.mat_t(blk.ff.net[2]), lora=[blk.ff.net[2]])
            d_h = self.buf(M, hpre.shape[1])
            ops.geglu_bwd(hpre, d_g, d_h)
            self.pool.put(d_g, hpre)
            d_ln3 = self.lin_b(d_h, pk._memo(("geglu_t", id(proj.proj)), lambda: wg.t().contiguous()), lora=[proj.proj])
            self.pool.put(d_h)
            d_y2 = ln_b(blk.norm3, y2, d_ln3, dy3)
            self.pool.put(d_ln3, dy3, y2)
            d_o2 = self.lin_b(d_y2, pk.mat_t(a2.to_out[0]), lora=[a2.to_out[0]])
            d_ln2 = attn2_b(d_o2)
            d_y1 = ln_b(blk.norm2, y1, d_ln2, d_y2)
            self.pool.put(d_ln2, d_y2, y1)
            d_o1 = self.lin_b(d_y1, pk.mat_t(a1.to_out[0]), lora=[a1.to_out[0]])
            d_ln1 = attn1_b(d_o1)
            d_y = ln_b(blk.norm1, y, d_ln1, d_y1)
            self.pool.put(d_ln1, d_y1, y)
            return d_y

        return y3, bwd
