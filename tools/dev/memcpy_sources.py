"""Which CPU ops issue the device copies (hipMemcpy* -> __amd_rocclr_copyBuffer) of one training step"""
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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU:
        continue
    name = ev.name
    if "Memcpy" in name or "memcpy" in name or "copyBuffer" in name or "Memset" in name or "fillBuffer" in name:
        cnt[name[:80]] += 1
print("device-side copy / fill events in one step:")
for k, v in cnt.most_common():
    print(f"{v:4d}  {k}")
cpu = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and any("emcpy" in k.name or "copyBuffer" in k.name for k in ev.kernels):
        if ev.cpu_parent is None or not ev.cpu_parent.name.startswith("aten::"):
            cpu[ev.name] += sum(1 for k in ev.kernels if "emcpy" in k.name or "copyBuffer" in k.name)
print("by top-level aten op:")
for k, v in cpu.most_common():
    print(f"{v:4d}  {k}")
