"""Which lines of this package issue the framework (aten) kernels of one training step: a TorchDispatchMode logs every
non-view aten call on device tensors with the innermost simvg_amd / bench frame (autograd-engine accumulations show up
under the `.backward()` call site)."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from simvg_amd.models import build_model
from simvg_amd.core import build_optimizer
from simvg_amd.graphs import training_stream

VIEWS = {"view", "reshape", "slice", "select", "expand", "t", "transpose", "detach", "alias", "as_strided", "unsqueeze",
         "squeeze", "permute", "empty", "empty_like", "_unsafe_view", "split", "unbind", "empty_strided", "new_empty",
         "is_same_size", "_reshape_alias", "unsafe_split", "chunk", "narrow", "view_as", "lift_fresh", "split_with_sizes",
         "new_empty_strided", "_local_scalar_dense", "is_pinned", "stride", "size", "sym_size", "numel", "dim", "is_contiguous"}


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.overloadpacket.__name__
        if name in VIEWS:
            return out
        flat = [a for a in torch.utils._pytree.tree_leaves((args, kwargs, out)) if isinstance(a, torch.Tensor)]
        if not any(t.is_cuda for t in flat):
            return out
        site = "?"
        for fr in reversed(traceback.extract_stack()[:-1]):
            if ("simvg_amd" in fr.filename or fr.filename.endswith("aten_sites.py")) and "aten_sites.py:__torch" not in fr.filename:
                if fr.name == "__torch_dispatch__":
                    continue
                site = f"{fr.filename.split('/root/repo/')[-1]}:{fr.lineno} {fr.name}"
                break
        self.sites[(name, site)] += 1
        return out


dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = build_model(bench.model_cfg()).to(dev).train()
model.vis_enc._ensure_engine(dev)
named = list(model.named_parameters())
groups = [{"params": [p for n, p in named if "vis_enc" in n], "lr": 5e-5}, {"params": [], "lr": 5e-4},
          {"params": [p for n, p in named if "vis_enc" not in n], "lr": 5e-4}]
opt = build_optimizer(dict(type="Adam", lr=5e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0, amsgrad=True), groups, model=model)
b = bench.synthetic_batch(64, 1, dev)
torch.autograd.set_multithreading_enabled(False)


def step():
    losses, _ = model(b["img"], b["ref_expr_inds"], b["img_metas"], return_loss=True, text_attention_mask=b["text_attention_mask"],
                      gt_bbox=b["gt_bbox"], rescale=False)
    opt.zero_grad()
    losses["loss_total"].backward()
    opt.clip_grad_norm(0.15)
    opt.step()


with training_stream(dev):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    log = Log()
    with log:
        step()
    torch.cuda.synchronize()
print("aten calls on device tensors in one step:", sum(log.sites.values()))
for (name, site), n in sorted(log.sites.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"{n:4d}  {name:28s} {site}")
