"""Optimizer glue: the reference's Adam(amsgrad) + global-norm clip (tools/train.py:78-94, apis/train.py:75-83,
core/optimizer.py:52-68) stay in PyTorch-ROCm (north_star: "outer training loop only"), but are run over the FLAT
arenas so the step is a few fused multi-tensor kernels instead of ~600 small ones.  Element-wise Adam over a flat
view is bit-identical to per-tensor Adam; the global gradient norm is the same number."""
import torch


def build_adam(model, lr=5e-4, lr_vis_enc=None, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.0, amsgrad=True):
    """Param groups split by name like the reference ("vis_enc" -> lr/10, rest -> lr)."""
    lr_vis_enc = lr / 10.0 if lr_vis_enc is None else lr_vis_enc
    enc = [p for n, p in model.named_parameters() if "vis_enc" in n and p.requires_grad]
    rest = [p for n, p in model.named_parameters() if "vis_enc" not in n and p.requires_grad]
    groups = [{"params": enc, "lr": lr_vis_enc}, {"params": rest, "lr": lr}]
    return torch.optim.Adam(groups, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)


class FlatAdam:
    """Adam(amsgrad) over the encoder arena as ONE tensor + the head parameters; same math as build_adam()."""

    def __init__(self, model, lr=5e-4, lr_vis_enc=None, betas=(0.9, 0.98), eps=1e-9, amsgrad=True, max_norm=0.15):
        self.model, self.max_norm = model, max_norm
        lr_vis_enc = lr / 10.0 if lr_vis_enc is None else lr_vis_enc
        enc = model.vis_enc
        if enc._arena is None:
            raise RuntimeError("run one forward (or call model.vis_enc._ensure_engine(device)) before building FlatAdam")
        self.arena = enc._arena
        self.flat = torch.nn.Parameter(self.arena.flat)
        self.flat.grad = self.arena.flat_grad
        self.rest = [p for n, p in model.named_parameters() if "vis_enc" not in n and p.requires_grad]
        groups = [{"params": [self.flat], "lr": lr_vis_enc}, {"params": self.rest, "lr": lr}]
        try:      # single-pass multi-tensor kernel (p, g, m, v, vmax read once) instead of ~10 foreach passes
            self.opt = torch.optim.Adam(groups, lr=lr, betas=betas, eps=eps, amsgrad=amsgrad, fused=True)
        except (RuntimeError, ValueError):
            self.opt = torch.optim.Adam(groups, lr=lr, betas=betas, eps=eps, amsgrad=amsgrad)

    def zero_grad(self):
        for p in self.rest:
            p.grad = None
        for p in self.arena.params.values():
            p.grad = None      # the arena re-attaches (and zeroes) on the next backward

    def step(self):
        self.flat.grad = self.arena.flat_grad
        params = [self.flat] + [p for p in self.rest if p.grad is not None]
        norm = torch.nn.utils.clip_grad_norm_(params, self.max_norm) if self.max_norm else None
        self.opt.step()
        return norm
