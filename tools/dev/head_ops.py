"""Attribute the head's small PyTorch glue launches (fills / copies / adds / ...) to source lines.

Runs two eager training steps of the full model (the second one logged) under a TorchDispatchMode that counts every
aten op that launches a kernel, keyed by (op, innermost simvg_amd frame).  Ops issued by the autograd engine for built-in
nodes (AddBackward, AccumulateGrad ...) have no Python frame and are shown as "<autograd>".
    python tools/dev/head_ops.py [--batch 64]
"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                                                   # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode     # noqa: E402

SKIP = {"aten.view", "aten._unsafe_view", "aten.detach", "aten.alias", "aten.slice", "aten.select", "aten.expand",
        "aten.unsqueeze", "aten.squeeze", "aten.t", "aten.transpose", "aten.permute", "aten.as_strided", "aten.empty",
        "aten.empty_like", "aten.empty_strided", "aten.reshape", "aten.unbind", "aten.split", "aten.is_same_size",
        "aten.new_empty", "aten.lift_fresh", "aten.unfold", "aten.split_with_sizes", "aten.view_as_real",
        "aten._local_scalar_dense", "aten.is_pinned", "aten.set_", "aten.resize_", "aten.new_empty_strided"}


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.counts = collections.Counter()
        self.on = False

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if self.on:
            name = str(func).rsplit(".", 1)[0] if str(func).count(".") > 1 else str(func)
            if name not in SKIP:
                where = "<autograd>"
                for fr in reversed(traceback.extract_stack(limit=14)[:-1]):
                    if "simvg_amd" in fr.filename and "head_ops" not in fr.filename:
                        where = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}"
                        break
                self.counts[(name, where)] += 1
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    os.environ["SIMVG_HEAD_GRAPH"] = "0"
    import bench
    from simvg_amd.models import build_model
    device = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = build_model(bench.model_cfg(1, "base")).to(device).train()
    batch = bench.synthetic_batch(a.batch, 1000, device)

    def step():
        losses, _ = model(batch["img"], batch["ref_expr_inds"], batch["img_metas"], return_loss=True,
                          text_attention_mask=batch["text_attention_mask"], gt_bbox=batch["gt_bbox"], rescale=False)
        for p in model.parameters():
            p.grad = None
        losses["loss_total"].backward()

    step()
    c = Counter()
    with c:
        step()
        c.on = True
        step()
        c.on = False
    torch.cuda.synchronize()
    by_op = collections.Counter()
    for (op, where), n in c.counts.items():
        by_op[op] += n
    print("== per op")
    for op, n in by_op.most_common():
        print(f"{n:5d} {op}")
    print("== per (op, site)")
    for (op, where), n in sorted(c.counts.items(), key=lambda kv: -kv[1]):
        print(f"{n:5d} {op:28s} {where}")


if __name__ == "__main__":
    main()
