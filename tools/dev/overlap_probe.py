"""Upper bound of what overlapping the optimizer tail (norm + fused Adam) with the next step's forward could give:
the optimizer is issued on a side stream and the next forward does NOT wait for it (results are garbage -- timing only).
    python tools/dev/overlap_probe.py [--steps 60]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch          # noqa: E402
import bench          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    a = ap.parse_args()
    from simvg_amd.models import build_model
    from simvg_amd.core import build_optimizer
    from simvg_amd.graphs import train_stream
    device = torch.device("cuda", 0)
    torch.manual_seed(1234)
    model = build_model(bench.model_cfg(1, "base")).to(device).train()
    batch = bench.synthetic_batch(64, 1000, device)
    model.vis_enc._ensure_engine(device)
    named = list(model.named_parameters())
    groups = [{"params": [p for n, p in named if "vis_enc" in n], "lr": 5e-5},
              {"params": [p for n, p in named if "vis_enc" not in n], "lr": 5e-4}]
    opt = build_optimizer(dict(type="Adam", lr=5e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0, amsgrad=True), groups, model=model)
    main_s = train_stream(device)
    side = torch.cuda.Stream(device=device)

    def step(overlap):
        losses, _ = model(batch["img"], batch["ref_expr_inds"], batch["img_metas"], return_loss=True,
                          text_attention_mask=batch["text_attention_mask"], gt_bbox=batch["gt_bbox"], rescale=False)
        opt.zero_grad()
        losses["loss_total"].backward()
        if overlap:
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                opt.clip_grad_norm(0.15)
                opt.step()
        else:
            opt.clip_grad_norm(0.15)
            opt.step()

    with torch.cuda.stream(main_s):
        for mode in (False, True, False, True):
            for _ in range(12):
                step(mode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step(mode)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.steps
            print(f"optimizer on side stream, next forward not waiting = {mode}: {dt * 1e3:.3f} ms/step  {64 / dt:.1f} pairs/s", flush=True)


if __name__ == "__main__":
    main()
