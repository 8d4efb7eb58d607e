"""The encoder arena's Adam launch with the training step's REAL state (after a few bench steps), timed back to back:
separates "the data" from "the neighbourhood" as the reason for in-situ 1.2 ms against 0.75 ms on synthetic arrays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from simvg_amd import hip_ops as ops
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
batches = [bench.synthetic_batch(64, 100 + i, dev) for i in range(4)]
with training_stream(dev):
    for i in range(6):
        b = batches[i % 4]
        losses, _ = model(b["img"], b["ref_expr_inds"], b["img_metas"], return_loss=True, text_attention_mask=b["text_attention_mask"],
                          gt_bbox=b["gt_bbox"], rescale=False)
        opt.zero_grad()
        losses["loss_total"].backward()
        opt.clip_grad_norm(0.15)
        if i < 5:
            opt.step()
    torch.cuda.synchronize()
    A = opt.arena
    st = opt.state[opt.flat]
    g = A.flat_grad
    n = g.numel()
    print("arena elements:", n, " zero gradient fraction:", float((g == 0).float().mean()), " zero exp_avg fraction:",
          float((st["exp_avg"] == 0).float().mean()))
    tn = torch.full((1,), 1e9, device=dev)

    def adam(gg=g, m=st["exp_avg"], v=st["exp_avg_sq"], vm=st["max_exp_avg_sq"], p=A.flat):
        ops.adam_step(p, gg, m, v, vm, 0.0, 0.9, 0.9, 0.98, 1e-9, total_norm=tn, max_norm=0.15)

    def timed(fn, reps=10):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    print(f"adam on the real state, back to back: {timed(adam):8.1f} us")
    cp = [t.clone() for t in (A.flat, g, st['exp_avg'], st['exp_avg_sq'], st['max_exp_avg_sq'])]
    print(f"adam on clones of the real state:     {timed(lambda: adam(cp[1], cp[2], cp[3], cp[4], cp[0])):8.1f} us")
    rnd = [torch.randn(n, device=dev).abs_() for _ in range(5)]
    print(f"adam on dense random arrays, same n:  {timed(lambda: adam(rnd[1], rnd[2], rnd[3], rnd[4], rnd[0])):8.1f} us")
    # element-wise zero structure of the real gradient: fraction of float4 chunks that are all-zero in g, m and v
    z4 = ((g.view(-1, 4) == 0).all(1) & (st["exp_avg"].view(-1, 4) == 0).all(1) & (st["exp_avg_sq"].view(-1, 4) == 0).all(1))
    print("all-zero float4 chunks (skipped):", float(z4.float().mean()))
