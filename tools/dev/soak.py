"""Sustained training run on the bench configuration (ViT-B/32, B = 64, 8 rotating synthetic batches): loss trajectory, step
time drift and the gradient scale -- `python tools/dev/soak.py [steps]`; `SOAK_VIT=large SOAK_BATCH=32 SOAK_QUERIES=10` for the other
BASELINE configurations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from simvg_amd.models import build_model
from simvg_amd.core import build_optimizer
from simvg_amd.graphs import training_stream

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
VIT, BATCH, NQ = os.environ.get("SOAK_VIT", "base"), int(os.environ.get("SOAK_BATCH", 64)), int(os.environ.get("SOAK_QUERIES", 1))
model = build_model(bench.model_cfg(NQ, VIT)).to(dev).train()
model.vis_enc._ensure_engine(dev)
named = list(model.named_parameters())
groups = [{"params": [p for n, p in named if "vis_enc" in n], "lr": 5e-5}, {"params": [], "lr": 5e-4},
          {"params": [p for n, p in named if "vis_enc" not in n], "lr": 5e-4}]
opt = build_optimizer(dict(type="Adam", lr=5e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0, amsgrad=True), groups, model=model)
batches = [bench.synthetic_batch(BATCH, 1000 + i, dev) for i in range(8)]
hist, t_hist = [], []
with training_stream(dev):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        b = batches[i % 8]
        losses, _ = model(b["img"], b["ref_expr_inds"], b["img_metas"], return_loss=True, text_attention_mask=b["text_attention_mask"],
                          gt_bbox=b["gt_bbox"], rescale=False)
        opt.zero_grad()
        losses["loss_total"].backward()
        norm = opt.clip_grad_norm(0.15)
        opt.step()
        if i % 100 == 99 or i == 0:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            hist.append((i + 1, float(losses["loss_total"]), float(norm), model.vis_enc._scale_tracker.scale if hasattr(model.vis_enc._scale_tracker, "scale") else None))
            t_hist.append((i + 1, t1))
            print(f"step {i + 1:5d}  loss {hist[-1][1]:8.4f}  grad norm {hist[-1][2]:9.4f}  ms/step so far {(t1 - t0) / (i + 1) * 1e3:6.2f}", flush=True)
finite = all(torch.isfinite(p).all() for p in model.parameters())
print("all parameters finite:", bool(finite), " loss first/last:", hist[0][1], hist[-1][1])
