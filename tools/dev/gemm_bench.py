"""GEMM micro-benchmark for profiling: the four encoder shapes at B=64 (M = 26944, split 25664)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops

M, SPLIT = 26944, 25664
shapes = [("qkv", 2304, 768), ("out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
which = sys.argv[2] if len(sys.argv) > 2 else "nt"
only = sys.argv[3] if len(sys.argv) > 3 else None
dev = "cuda"
for name, N, K in shapes:
    if only and name != only:
        continue
    a = (torch.randn(M, K, device=dev) * 1.0).to(ops.LP())
    w = (torch.randn(2, N, K, device=dev) * K ** -0.5).to(ops.LP())
    bias = torch.randn(2, N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=ops.LP())
    dy = (torch.randn(M, N, device=dev)).to(ops.LP())
    dw = torch.zeros(2, N, K, device=dev)
    for _ in range(30):
        if which == "nt":
            ops.gemm_nt(a, w, bias=bias, out=out, split=SPLIT)
        else:
            ops.gemm_tn(dy, a, dw, split=SPLIT)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        if which == "nt":
            ops.gemm_nt(a, w, bias=bias, out=out, split=SPLIT)
        else:
            ops.gemm_tn(dy, a, dw, split=SPLIT)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{which} {name:4s} M={M} N={N} K={K}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
