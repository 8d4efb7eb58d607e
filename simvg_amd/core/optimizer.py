"""Optimizer registry -- mirror of the reference's `simvg/core/optimizer.py:1-87` (mmcv Registry wrapping
torch.optim.{SGD, RMSprop, Adam, AdamW}; `build_optimizer(cfg, params)` fills `params` as a default argument).

MI355X-first addition: `FlatAdam`, the same Adam(amsgrad) arithmetic run over the encoder's FLAT parameter / gradient
arenas (one tensor) plus the head's parameters, as one fused multi-tensor launch instead of ~600 per-tensor updates.
Element-wise Adam over a flat view is bit-identical to per-tensor Adam.  `build_optimizer(cfg, params, model=model)`
selects it for `type="Adam"` when the model carries arenas; the param groups keep the reference's order and learning
rates (tools/train.py:78-94: vis_enc -> lr_vis_enc, lan_enc -> lr_lan_enc, rest -> lr), so schedulers and the
`lr:{optimizer.param_groups[0]['lr']}` log field behave as in the reference."""
import os
import warnings

import torch

from ..models.builder import Registry

OPTIMIZERS = Registry("OPTIMIZERS")


@OPTIMIZERS.register_module()
class SGD(torch.optim.SGD):
    def __init__(self, params, lr, momentum=0, dampening=0, weight_decay=0, nesterov=False):
        super().__init__(params, lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=dampening,
                         nesterov=nesterov)


@OPTIMIZERS.register_module()
class RMSProp(torch.optim.RMSprop):
    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0, momentum=0, centered=False):
        super().__init__(params, lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay, momentum=momentum,
                         centered=centered)


@OPTIMIZERS.register_module()
class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        super().__init__(params, lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)


@OPTIMIZERS.register_module()
class AdamW(torch.optim.AdamW):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        super().__init__(params, lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)


class _RestArena:
    """The parameters OUTSIDE the encoder arena (the head's ~120 tensors) as views into one flat fp32 buffer, with a flat
    gradient buffer of the same layout that is filled from the autograd-produced `.grad`s by ONE multi-tensor copy per
    step.  Lets the clip norm and Adam(amsgrad) of those parameters be the same two HIP launches as the encoder's instead
    of the framework's foreach / fused-Adam chain (13 launches and ~2.5 ms of host time per step)."""

    ALIGN = 64

    def __init__(self, params):
        self.params = list(params)
        device = self.params[0].device
        off, self.offsets = 0, []
        for p in self.params:
            off = (off + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            self.offsets.append(off)
            off += p.numel()
        self.total = (off + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.flat = torch.zeros(self.total, device=device, dtype=torch.float32)
        self.flat_grad = torch.zeros(self.total, device=device, dtype=torch.float32)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                v = self.flat[o:o + p.numel()].view(p.shape)
                v.copy_(p.detach())
                p.data = v
        self.grad_views = [self.flat_grad[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, self.offsets)]

    def intact(self):
        base = self.flat.untyped_storage().data_ptr()
        ps = (self.params[0], self.params[len(self.params) // 2], self.params[-1])
        return all(p.data.untyped_storage().data_ptr() == base for p in ps)

    def gather_grads(self, rebind=False, check=True):
        """autograd's per-tensor gradients -> the flat gradient buffer.  A parameter whose `.grad` is None contributes
        zeros: per-tensor Adam SKIPS such a parameter (no moment decay, no step count, no state), and the fused kernel does
        exactly that for an element whose gradient and moments are all zero (the update is exactly zero and nothing is
        written) -- i.e. for a parameter that NEVER receives a gradient (the token branch under branch_loss_weight=
        {"decoder": w}, `mask_token`).  A parameter graded on SOME steps only departs from torch.optim.Adam on the steps
        without a gradient (its moments decay with a zero gradient and its bias correction uses the flat tensor's step
        count).  That is not the reference optimizer's arithmetic, so a changed set RAISES (round 5; no reference config
        produces one: every head node is in the graph on every step).  SIMVG_ALLOW_GRADED_SET_CHANGE=1 opts into
        warn-and-continue (the update stays finite and well defined); `optimizer_config.flat=False` is the exact alternative.

        A `.grad` that already IS its slice of the flat buffer is left alone: after the gradient exchange (`dist.GradReducer`
        gathers early, all-reduces the flat buffer in place and re-points every `.grad` at its slice) the optimizer's own gather
        copies nothing.  rebind=True re-points the gathered `.grad`s right away; `last_gathered` = [(slice, parameter)] of this call.
        check=False (the reducer's EARLY gather, from inside the backward): the has-a-gradient set is neither recorded nor
        compared -- a head gradient may still arrive after it (the reducer's late pass copies it into its slice); only the
        optimizer's own gather, which sees the step's final set, does the bookkeeping.
        -> number of parameters with a gradient"""
        mask = tuple(p.grad is not None for p in self.params)
        if not check:
            pass
        elif getattr(self, "_graded", None) is None:
            self._graded = mask
        elif mask != self._graded:
            changed = [i for i, (a, b) in enumerate(zip(mask, self._graded)) if a != b]
            msg = (f"FlatAdam: {len(changed)} parameter(s) changed between 'has a gradient' and 'has none' "
                   "(first: index %d, shape %s); on steps without a gradient their moments decay as if it were zero, "
                   "unlike per-tensor torch Adam which skips them" % (changed[0], tuple(self.params[changed[0]].shape)))
            if os.environ.get("SIMVG_ALLOW_GRADED_SET_CHANGE") != "1":
                raise RuntimeError(msg + "; use optimizer_config.flat=False (per-tensor Adam) for such a model, or "
                                   "SIMVG_ALLOW_GRADED_SET_CHANGE=1 to continue with the flat update")
            warnings.warn(msg, RuntimeWarning, stacklevel=2)
            self._graded = mask
        graded = sum(mask)
        have = [(v, p) for v, p in zip(self.grad_views, self.params) if p.grad is not None and p.grad is not v]
        if graded < len(self.params):
            # (before the early return below: with NO gradient at all the buffer must not keep the previous step's values)
            torch._foreach_zero_([v for v, p in zip(self.grad_views, self.params) if p.grad is None])
        if not have:
            self.last_gathered = []
            return graded              # every gradient already lives in its slice (gathered earlier in this step)
        torch._foreach_copy_([v for v, _ in have], [p.grad for _, p in have])
        self.last_gathered = have
        if rebind:
            for v, p in have:
                p.grad = v
        return graded


@OPTIMIZERS.register_module()
class FlatAdam(torch.optim.Adam):
    """Adam over [encoder arena as ONE tensor] + [every other parameter as ONE tensor].  `params` are the reference-style
    groups (lists of the model's nn.Parameters with an `lr` each); the group whose parameters all live in the encoder arena
    is replaced by the arena's flat tensor, the remaining non-empty groups by a flat tensor each (`_RestArena`: the
    parameters become views into it).  Each flat tensor is updated by ONE HIP kernel (`simvg_adam_step`: clip scale + Adam
    amsgrad, 36 B per parameter); the optimizer state keeps torch's keys (`step`, `exp_avg`, `exp_avg_sq`,
    `max_exp_avg_sq`) per flat tensor, so `state_dict()` round-trips.  `clip_grad_norm(max_norm)` is the global-norm clip of
    apis/train.py:81-82 over the same set of gradients: the coefficient is applied INSIDE the Adam kernels (the `.grad`
    tensors themselves stay unscaled)."""

    def __init__(self, params, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0:      # the fused kernel updates the WHOLE arena: parameters that never receive a gradient
            raise NotImplementedError(       # (vision_embed.mask_token) would decay, which per-tensor Adam does not do
                "FlatAdam implements the reference's setting weight_decay=0 (every config); use optimizer_config.flat=False "
                "for L2 weight decay")
        enc = getattr(model, "vis_enc", None)
        arena = getattr(enc, "_arena", None)
        if arena is None:
            raise RuntimeError("FlatAdam needs the encoder arena: move the model to its GPU and call "
                               "model.vis_enc._ensure_engine(device) (or run one forward) first")
        self.arena = arena
        self._model_enc = enc
        # modules that keep 16-bit copies of their weights: told after every step (the flat update does not move the
        # parameters' version counters, which is what those modules watch in eval mode)
        self._weight_watchers = [m.mark_weights_dirty for m in model.modules() if hasattr(m, "mark_weights_dirty")]
        in_arena = {id(p) for p in arena.params.values()}
        self._names = {id(p): n for n, p in model.named_parameters()}      # (layout signature: which parameter sits where)
        self.flat = torch.nn.Parameter(arena.flat)
        self.flat.grad = arena.flat_grad
        self._clip = None
        groups, self._rest, self._rest_arenas, used_flat = [], [], [], False
        for g in params:
            g = dict(g)
            ps = list(g["params"])
            if ps and all(id(p) in in_arena for p in ps):
                if used_flat:
                    raise ValueError("the encoder arena can belong to one param group only")
                frozen = [n for n, p in arena.params.items() if not p.requires_grad]
                if len(ps) + len(frozen) < len(arena.params) - len(arena.no_grad):
                    raise ValueError("FlatAdam updates the whole encoder arena; a partially selected encoder group "
                                     "(freeze_layer >= 0) must use type='Adam'")
                g["params"], used_flat = [self.flat], True
            else:
                if any(id(p) in in_arena for p in ps):
                    raise ValueError("a param group mixes encoder-arena and other parameters")
                if ps:
                    ra = _RestArena(ps)
                    ra.param = torch.nn.Parameter(ra.flat)
                    ra.param.grad = ra.flat_grad
                    self._rest_arenas.append(ra)
                    self._rest += ps
                    g["params"] = [ra.param]
            groups.append(g)      # empty groups (lan_enc) stay, so group indices match the reference's
        super().__init__(groups, lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)
        # the gradient exchange (dist.GradReducer) all-reduces these flat gradient buffers in place
        object.__setattr__(model, "_simvg_rest_arenas", self._rest_arenas)

    def layout_signature(self):
        """where every parameter sits inside the flat tensors (names are the model's, offsets the arenas'): the per-element moments
        of a saved state belong to THIS layout only"""
        import zlib
        text = ";".join(f"{n}@{o}:{self.arena.params[n].numel()}" for n, o in self.arena.offsets.items())
        for ra in self._rest_arenas:
            # (names too since round 6: two same-sized head parameters that swap places change the signature)
            text += "|" + ";".join(f"{self._names.get(id(p), '?')}@{o}:{p.numel()}" for p, o in zip(ra.params, ra.offsets))
        return zlib.crc32(text.encode())

    def state_dict(self):
        sd = super().state_dict()
        sd["simvg_layout"] = self.layout_signature()
        return sd

    def load_state_dict(self, state_dict):
        sd = dict(state_dict)
        sig = sd.pop("simvg_layout", None)
        if sig != self.layout_signature():
            # (round 5 moved the encoder's LayerNorm parameters behind the layers' Linears: a state written before has the same
            # size and a different order -- loading it would hand every element someone else's moments)
            # No migration path: a state written before the signature existed (or for another layout) carries per-element moments
            # whose owner cannot be recovered from the flat tensors alone.  Resume such a run with a fresh optimizer state (the
            # model weights load independently: checkpoint.load_checkpoint(model, path) without `resume`), or train with
            # optimizer_config.flat=False, whose state is torch.optim.Adam's per-parameter dict.
            raise ValueError("FlatAdam state was written for another arena layout (%r, this build: %r): resume with a fresh "
                             "optimizer state, or use optimizer_config.flat=False" % (sig, self.layout_signature()))
        super().load_state_dict(sd)

    def zero_grad(self, set_to_none=True):
        for p in self._rest:
            p.grad = None
        for p in self.arena.params.values():
            p.grad = None      # the arena re-attaches (and zeroes) its gradient views on the next backward
        self._clip = None
        self._gathered = False

    def _check_arena(self):
        enc_arena = getattr(getattr(self, "_model_enc", None), "_arena", None)
        if enc_arena is not self.arena or not self.arena.intact() or not all(ra.intact() for ra in self._rest_arenas):
            raise RuntimeError("the model re-created its parameter storage after this optimizer was built (model moved "
                               "to another device?): rebuild the optimizer")

    def _gather(self):
        if not getattr(self, "_gathered", False):
            for ra in self._rest_arenas:
                ra.gather_grads()
            self._gathered = True

    def clip_grad_norm(self, max_norm):
        """Global-norm clip (apis/train.py:81-82) without a host sync and without touching the gradients: the squared
        norm of every flat gradient buffer is reduced deterministically on the device, the clip coefficient is applied
        inside the Adam kernels of the following step().  Returns the total norm (device scalar)."""
        from .. import hip_ops as ops
        self._check_arena()
        self._gather()
        sq = torch.zeros(1, device=self.arena.flat.device, dtype=torch.float32)
        ops.sumsq_accum(self.arena.flat_grad, sq)
        for ra in self._rest_arenas:
            ops.sumsq_accum(ra.flat_grad, sq)
        total = sq.sqrt_()
        self._clip = (total, float(max_norm))
        return total.reshape(())

    def _flat_step(self, flat, grad, group):
        from .. import hip_ops as ops
        st = self.state[flat]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(flat.data)
            st["exp_avg_sq"] = torch.zeros_like(flat.data)
            if group["amsgrad"]:
                st["max_exp_avg_sq"] = torch.zeros_like(flat.data)
        if st["step"].is_cuda:      # load_state_dict may move `step` to the parameter's device: bring the counter home once,
            st["step"] = st["step"].cpu()     # or every step would synchronise on int(...)
        st["step"] += 1
        t = int(st["step"])
        b1, b2 = group["betas"]
        lr = float(group["lr"])
        total, max_norm = self._clip if self._clip is not None else (None, 0.0)
        ops.adam_step(flat.data, grad, st["exp_avg"], st["exp_avg_sq"], st.get("max_exp_avg_sq"),
                      lr / (1.0 - b1 ** t), (1.0 - b2 ** t) ** 0.5, b1, b2, group["eps"], group["weight_decay"],
                      total_norm=total, max_norm=max_norm)

    def step(self, closure=None):
        self._check_arena()
        self._gather()
        for group in self.param_groups:
            for p in group["params"]:
                if p is self.flat:
                    self._flat_step(p, self.arena.flat_grad, group)
                else:
                    ra = next(r for r in self._rest_arenas if r.param is p)
                    self._flat_step(p, ra.flat_grad, group)
        self._clip = None
        self._gathered = False
        for mark in self._weight_watchers:
            mark()
        return None


def build_optimizer(cfg, params, model=None):
    """Reference signature `build_optimizer(cfg, params)`.  With `model=` given, `type="Adam"` resolves to FlatAdam
    when the model has its encoder arena (same arithmetic, one fused launch); `flat=False` in cfg opts out."""
    cfg = dict(cfg)
    flat = cfg.pop("flat", True)
    if model is not None and flat and cfg.get("type") == "Adam":
        enc = getattr(model, "vis_enc", None)
        arena = getattr(enc, "_arena", None)
        enc_all_trainable = arena is not None and all(
            p.requires_grad for n, p in arena.params.items() if n not in arena.no_grad)
        if enc_all_trainable:
            return OPTIMIZERS.build(dict(cfg, type="FlatAdam"), default_args=dict(params=params, model=model))
    return OPTIMIZERS.build(cfg, default_args=dict(params=params))
