"""Do all head gradients make the early packed message of GradReducer with the eager head?  (1 rank, RCCL)
    python tools/dev/late_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
import bench
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ["SIMVG_FORCE_REDUCE"] = "1"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from simvg_amd.models import build_model
from simvg_amd.dist import GradReducer
from simvg_amd.graphs import train_stream
dev = torch.device("cuda", 0)
model = build_model(bench.model_cfg(1, "base")).to(dev).train()
batch = bench.synthetic_batch(8, 1000, dev)
red = GradReducer(model)
with torch.cuda.stream(train_stream(dev)):
    for i in range(3):
        losses, _ = model(batch["img"], batch["ref_expr_inds"], batch["img_metas"], return_loss=True,
                          text_attention_mask=batch["text_attention_mask"], gt_bbox=batch["gt_bbox"], rescale=False)
        for p in model.parameters():
            p.grad = None
        red.begin()
        losses["loss_total"].backward()
        n_head = sum(1 for n, p in model.named_parameters() if not n.startswith("vis_enc.") and p.grad is not None)
        red.finish()
        print("step", i, "head params with grad", n_head, "late", red.last_late, flush=True)
torch.cuda.synchronize()
dist.destroy_process_group()
