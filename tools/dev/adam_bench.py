"""Optimizer micro-benchmark on an arena of ViT-B / ViT-L size: python tools/dev/adam_bench.py [n_millions]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops
n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 141_000_000
n -= n % 4
dev = "cuda"
p, g = torch.randn(n, device=dev), torch.randn(n, device=dev) * 1e-3
m, v, vm = torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
sq = torch.zeros(1, device=dev)
def run():
    sq.zero_()
    ops.sumsq_accum(g, sq)
    ops.adam_step(p, g, m, v, vm, 1e-4, 1.0, 0.9, 0.98, 1e-9, 0.0, total_norm=sq.sqrt(), max_norm=0.15)
for _ in range(5): run()
torch.cuda.synchronize()
for name, fn in (("sumsq", lambda: ops.sumsq_accum(g, sq)),
                 ("adam", lambda: ops.adam_step(p, g, m, v, vm, 1e-4, 1.0, 0.9, 0.98, 1e-9, 0.0, total_norm=sq, max_norm=0.15))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    nb = n * (4 if name == "sumsq" else 36)
    print(f"{name:6s} n={n / 1e6:.0f} M: {us:8.1f} us  {nb / us / 1e6:6.2f} TB/s", flush=True)
