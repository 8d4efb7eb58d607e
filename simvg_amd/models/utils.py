"""`ExponentialMovingAverage` -- mirror of the reference's `simvg/models/utils.py:132-186` (used by
tools/train.py:104,133-138 and tools/test.py:69-85), as fused device lerps.

Reference semantics kept exactly:
  * `shadow` is a dict over `model.state_dict()` keys (parameters AND buffers), cloned at construction; it is what
    `save_checkpoint` stores as `ema_state_dict` and what `load_checkpoint` assigns back;
  * `update_params()`: decay = min(alpha, (step+1)/(step+10)); shadow = decay*shadow + (1-decay)*state for every
    parameter, and for every buffer when `buffer_ema` (else the buffer is copied); then step += 1;
  * `apply_shadow()` backs the live state up and loads the shadow (strict); `restore()` loads the backup.

MI355X-first: the encoder's parameters are views of ONE flat arena, so their shadows are views of one flat shadow
tensor and the update of ~600 tensors is a single `lerp_` over the arena plus one `_foreach_lerp_` over the head's
tensors -- no state_dict() walk, no per-tensor temporaries (the reference allocates 3 temporaries per tensor per step).
`shadow.lerp_(state, 1-decay)` computes shadow + (1-decay)*(state-shadow), the same value as
decay*shadow + (1-decay)*state up to one rounding."""
import torch


class ExponentialMovingAverage(object):
    def __init__(self, model, alpha, buffer_ema=True):
        self.step = 0
        self.model = model
        self.alpha = alpha
        self.buffer_ema = buffer_ema
        self.param_keys = [k for k, _ in self.model.named_parameters()]
        self.buffer_keys = [k for k, _ in self.model.named_buffers()]
        self.backup = {}
        self._flat_shadow = None
        self._shadow = None
        self.shadow = self.get_model_state()

    # `shadow` is assignable (load_checkpoint does `model_ema.shadow = ckpt["ema_state_dict"]`): re-pack on assignment
    @property
    def shadow(self):
        return self._shadow

    @shadow.setter
    def shadow(self, state):
        self._pack(state)

    def _arena(self):
        enc = getattr(getattr(self.model, "module", self.model), "vis_enc", None)
        arena = getattr(enc, "_arena", None)
        return arena if (arena is not None and arena.intact()) else None

    def _pack(self, state):
        """Lay the shadow out like the live state: arena-resident tensors become views of one flat shadow."""
        live = self.model.state_dict()
        arena = self._arena()
        shadow, flat = {}, None
        by_ptr = {}
        if arena is not None:
            flat = arena.flat.detach().clone()
            base, esz = arena.flat.data_ptr(), arena.flat.element_size()
            for k, v in live.items():
                off = v.data_ptr() - base
                if v.dtype == arena.flat.dtype and 0 <= off < arena.flat.numel() * esz and v.is_contiguous():
                    by_ptr[k] = off // esz
        for k, v in state.items():
            ref = live.get(k)
            if k in by_ptr and ref is not None and tuple(v.shape) == tuple(ref.shape):
                view = flat[by_ptr[k]: by_ptr[k] + ref.numel()].view(ref.shape)
                view.copy_(v.detach().to(view.device, view.dtype))
                shadow[k] = view
            else:
                dev = ref.device if ref is not None else v.device
                shadow[k] = v.detach().clone().to(dev)
        self._shadow, self._flat_shadow = shadow, flat
        self._flat_keys = set(by_ptr) & set(shadow)

    def update_params(self):
        decay = min(self.alpha, (self.step + 1) / (self.step + 10))
        w = 1.0 - decay
        state = self.model.state_dict()
        arena = self._arena()
        flat_ok = self._flat_shadow is not None and arena is not None and \
            self._flat_shadow.numel() == arena.flat.numel()
        if flat_ok:
            self._flat_shadow.lerp_(arena.flat.detach(), w)          # every encoder parameter in one launch
        dst, src = [], []
        for name in self.param_keys + (self.buffer_keys if self.buffer_ema else []):
            if name not in state:
                continue                                             # non-persistent buffer
            if name not in self._shadow:
                self._shadow[name] = state[name].detach().clone()
            elif flat_ok and name in self._flat_keys:
                continue
            elif self._shadow[name].is_floating_point():
                dst.append(self._shadow[name]); src.append(state[name].detach())
            else:   # integer buffers: the reference's copy_ of a float expression truncates toward the old value
                self._shadow[name].copy_(decay * self._shadow[name] + w * state[name])
        if dst:
            torch._foreach_lerp_(dst, src, w)
        if not self.buffer_ema:
            for name in self.buffer_keys:
                if name not in state:
                    continue
                if name not in self._shadow:
                    self._shadow[name] = state[name].detach().clone()
                else:
                    self._shadow[name].copy_(state[name])
        self.step += 1

    def apply_shadow(self):
        self.backup = self.get_model_state()
        self.model.load_state_dict(self._shadow, strict=True)
        self._weights_changed()

    def restore(self):
        self.model.load_state_dict(self.backup, strict=True)
        self._weights_changed()

    def _weights_changed(self):
        # the encoder keeps bf16 copies of its weights; make the next forward (eval included) refresh them
        for m in getattr(self.model, "module", self.model).modules():
            if hasattr(m, "mark_weights_dirty"):
                m.mark_weights_dirty()

    def get_model_state(self):
        return {k: v.clone().detach() for k, v in self.model.state_dict().items()}
