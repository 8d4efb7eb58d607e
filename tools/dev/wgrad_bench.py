"""wgrad micro-benchmark on the encoder's shapes (GPU box): python tools/dev/wgrad_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops
dev = "cuda"
LP = ops.LP()
B = int(os.environ.get("B", 64))
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
M, SPLIT = B * 421, B * 401
D = int(os.environ.get("D", 768))          # D=1024 B=32: the ViT-L shapes
for name, N, K in [("qkv", 3 * D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D), ("out", D, D)]:
    dy = torch.randn(M, N, device=dev).to(LP)
    x = torch.randn(M, K, device=dev).to(LP)
    dw = torch.zeros(2, N, K, device=dev)
    db = torch.zeros(2, N, device=dev)
    for _ in range(30):
        ops.gemm_tn(dy, x, dw, split=SPLIT, db=db)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        ops.gemm_tn(dy, x, dw, split=SPLIT, db=db)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / REPS * 1e3
    print(f"wgrad {name:4s} [{M}x{N}x{K}] {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
