"""Which Python lines launch the remaining at:: / copy / fill kernels of a training step (torch.profiler, with_stack)"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from simvg_amd.models import build_model
from simvg_amd.core import build_optimizer
from simvg_amd.graphs import training_stream

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
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
ops = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.cpu_parent is None or (ev.cpu_parent is not None and not ev.cpu_parent.name.startswith("aten::")):
        if not ev.name.startswith("aten::"):
            continue
        kern = sum(1 for k in ev.kernels) if hasattr(ev, "kernels") else 0
        if kern == 0:
            continue
        frame = next((f for f in (ev.stack or []) if "simvg_amd" in f or "bench" in f), "?")
        ops[(ev.name, frame.split("/root/repo/")[-1][:90])] += kern
tot = sum(ops.values())
print("device launches issued by aten ops in one step:", tot)
for (name, frame), n in ops.most_common(80):
    print(f"{n:4d}  {name:32s} {frame}")
