"""cProfile of the training step's host side (autograd single-threaded so that the backward's Python shows up)"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
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
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        step()
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(os.environ.get("TOP", 40)))
