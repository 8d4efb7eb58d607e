"""Which 16-bit gradient scale the tracker settles on during bench-like training steps (B = 64, synthetic batches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from simvg_amd.models import build_model
from simvg_amd.core import build_optimizer
from simvg_amd import hip_ops as ops

dev = torch.device("cuda", 0)
torch.manual_seed(1234)
B = int(os.environ.get("B", 64))
model = build_model(bench.model_cfg()).to(dev).train()
model.vis_enc._ensure_engine(dev)
named = list(model.named_parameters())
groups = [{"params": [p for n, p in named if "vis_enc" in n], "lr": 5e-5}, {"params": [], "lr": 5e-4},
          {"params": [p for n, p in named if "vis_enc" not in n], "lr": 5e-4}]
opt = build_optimizer(dict(type="Adam", lr=5e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0, amsgrad=True), groups, model=model)
for i in range(12):
    b = bench.synthetic_batch(B, 1000 + i, dev)
    losses, _ = model(b["img"], b["ref_expr_inds"], b["img_metas"], return_loss=True, text_attention_mask=b["text_attention_mask"],
                      gt_bbox=b["gt_bbox"], rescale=False)
    opt.zero_grad()
    losses["loss_total"].backward()
    opt.clip_grad_norm(0.15)
    opt.step()
    torch.cuda.synchronize()
    print(f"step {i}: loss {float(losses['loss_total']):.3f}  grad scale 2^{int(torch.log2(torch.tensor(ops.grad_scale())))}  "
          f"max|dout| seen {float(model.vis_enc._scale_tracker._host[0]):.3e}")
