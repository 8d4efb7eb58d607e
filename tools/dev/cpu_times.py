"""Host-side cost of one training step, per section (no device synchronisation inside the loop): is the CPU ahead of the
GPU?  python tools/dev/cpu_times.py [--steps 50]"""
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
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--sync-before-backward", action="store_true", help="drain the GPU before backward(): replay host time = pure launch cost")
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
    replay_t = [0.0, 0]
    per_graph = {}
    orig_replay = torch.cuda.CUDAGraph.replay

    def timed_replay(self):
        t = time.perf_counter()
        orig_replay(self)
        dt = time.perf_counter() - t
        replay_t[0] += dt
        replay_t[1] += 1
        d = per_graph.setdefault(id(self), [0.0, 0])
        d[0] += dt
        d[1] += 1
    torch.cuda.CUDAGraph.replay = timed_replay
    names = ["forward", "zero_grad", "backward", "clip", "opt.step"]
    acc = [0.0] * len(names)

    def step(rec):
        t = [time.perf_counter()]
        losses, _ = model(batch["img"], batch["ref_expr_inds"], batch["img_metas"], return_loss=True,
                          text_attention_mask=batch["text_attention_mask"], gt_bbox=batch["gt_bbox"], rescale=False)
        t.append(time.perf_counter())
        opt.zero_grad()
        if a.sync_before_backward:
            torch.cuda.synchronize()
        t.append(time.perf_counter())
        losses["loss_total"].backward()
        t.append(time.perf_counter())
        opt.clip_grad_norm(0.15)
        t.append(time.perf_counter())
        opt.step()
        t.append(time.perf_counter())
        if rec:
            for i in range(len(names)):
                acc[i] += t[i + 1] - t[i]

    with torch.cuda.stream(train_stream(device)):
        for _ in range(12):
            step(False)
        torch.cuda.synchronize()
        replay_t[0], replay_t[1] = 0.0, 0
        per_graph.clear()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step(True)
        t_cpu = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
    print(f"wall {t_all / a.steps * 1e3:.2f} ms/step; host done after {t_cpu / a.steps * 1e3:.2f} ms/step")
    print(f"  hipGraph replay calls: {replay_t[1] / a.steps:.1f} per step, {replay_t[0] / max(replay_t[1], 1) * 1e3:.3f} ms of host time each")
    for k, (tt, n) in per_graph.items():
        print(f"    graph {k & 0xffff:04x}: {n / a.steps:.1f} replays per step, {tt / n * 1e3:.3f} ms of host time each")
    for n, v in zip(names, acc):
        print(f"  {n:10s} {v / a.steps * 1e3:7.3f} ms")


if __name__ == "__main__":
    main()
