"""Flat parameter / gradient arenas in HBM.

MI355X-first layout decision: every parameter of a model part (encoder, head) is a VIEW into one
contiguous fp32 buffer, and every `.grad` a view into a second buffer of the same layout.  That gives
 * fused operands without copies: q|k|v weights of both multiway experts are adjacent, so the QKV
   projection is ONE grouped GEMM over a [2, 3D, D] view, and its weight gradient is written by ONE
   wgrad kernel straight into the matching view of the gradient arena;
 * few, large RCCL all-reduces over contiguous slices (one per encoder layer, issued as soon as that
   layer's backward is done) instead of hundreds of per-tensor messages;
 * a single batched fp32->bf16 (+transpose) weight-prep launch per optimizer step.
The nn.Parameter objects keep the reference's names, so state_dict()/load_state_dict()/optimizers see
exactly the reference schema.
"""
import torch

ALIGN = 64  # elements (256 B)


class ParamArena:
    def __init__(self, named_params, groups, device, no_grad=()):
        """named_params: ordered dict name -> Parameter.  groups: list of (view_name, [param names], shape)
        describing fused views; parameters not named in any group are laid out afterwards in order."""
        self.device = torch.device(device)
        self.params = dict(named_params)
        self.no_grad = set(no_grad)     # parameters the path never differentiates (their .grad stays None)
        order, seen = [], set()
        self._group_spec = []
        for vname, names, shape in groups:
            for n in names:
                assert n in self.params, n
                assert n not in seen, f"{n} appears in two fused groups"
                seen.add(n)
            order.append(list(names))
            self._group_spec.append((vname, names, shape))
        for n in self.params:
            if n not in seen:
                order.append([n])
        self.offsets = {}
        off = 0
        self.chunk_bounds = []
        for chunk in order:
            off = (off + ALIGN - 1) // ALIGN * ALIGN
            start = off
            for n in chunk:
                self.offsets[n] = off
                off += self.params[n].numel()
            self.chunk_bounds.append((start, off))
        self.total = (off + ALIGN - 1) // ALIGN * ALIGN
        self.flat = torch.zeros(self.total, device=self.device, dtype=torch.float32)
        self.flat_grad = torch.zeros(self.total, device=self.device, dtype=torch.float32)
        with torch.no_grad():
            for n, p in self.params.items():
                v = self.flat[self.offsets[n]: self.offsets[n] + p.numel()].view(p.shape)
                v.copy_(p.detach().to(self.device, torch.float32))
                p.data = v
        self.views, self.grad_views = {}, {}
        for vname, names, shape in self._group_spec:
            s = self.offsets[names[0]]
            n = sum(self.params[k].numel() for k in names)
            self.views[vname] = self.flat[s:s + n].view(shape)
            self.grad_views[vname] = self.flat_grad[s:s + n].view(shape)
        self._grad_of = {n: self.flat_grad[self.offsets[n]: self.offsets[n] + p.numel()].view(p.shape)
                         for n, p in self.params.items()}
        self._param_list = list(self.params.values())

    def intact(self, thorough=False):
        """False if someone re-allocated the parameters (e.g. model.to(other_device)).  Called several times per step
        (engine, optimizer, EMA): a move re-allocates EVERY parameter, so the first, the middle and the last one are the
        per-step probe (walking all ~500 cost 0.5 ms per call); every 256th call and `thorough=True` check them all."""
        base = self.flat.untyped_storage().data_ptr()
        ps = self._param_list
        self._intact_calls = getattr(self, "_intact_calls", 0) + 1
        if not thorough and self._intact_calls % 256:
            ps = (ps[0], ps[len(ps) // 2], ps[-1])
        return all(p.data.untyped_storage().data_ptr() == base for p in ps)

    def grad(self, name):
        return self._grad_of[name]

    def slice_of(self, names):
        lo = min(self.offsets[n] for n in names)
        hi = max(self.offsets[n] + self.params[n].numel() for n in names)
        return lo, hi

    def complement_views(self, key, ranges):
        """views of the gradient arena OUTSIDE the given (offset, numel) ranges, cached under `key`"""
        cache = self.__dict__.setdefault("_complement", {})
        if key not in cache:
            views, pos = [], 0
            for off, n in sorted(ranges):
                if off > pos:
                    views.append(self.flat_grad[pos:off])
                pos = max(pos, off + n)
            if pos < self.total:
                views.append(self.flat_grad[pos:self.total])
            cache[key] = views
        return cache[key]

    def begin_backward(self, assigned=None):
        """Attach .grad views.  If any grad was dropped (zero_grad(set_to_none=True)) the arena is
        zeroed first; otherwise kernels keep accumulating (PyTorch's += semantics).
        assigned = (key, [(offset, numel), ...]): ranges this backward WRITES (first accumulation: `gemm_tn(assign=True)`) -- a fresh
        arena then zeroes only the rest, in one multi-tensor launch.  -> fresh"""
        fresh = False
        for n, p in self.params.items():
            if not p.requires_grad or n in self.no_grad:
                continue
            g = p.grad
            if g is None or g.data_ptr() != self._grad_of[n].data_ptr():
                fresh = True
                break
        if fresh:
            if assigned is None:
                self.flat_grad.zero_()
            else:
                torch._foreach_zero_(self.complement_views(*assigned))
            for n, p in self.params.items():
                if p.requires_grad and n not in self.no_grad:
                    p.grad = self._grad_of[n]
        return fresh
