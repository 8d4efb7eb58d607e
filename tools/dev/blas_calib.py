"""Calibration only (not a product path): what does the vendor library reach on the encoder GEMM shapes?"""
import torch
M = 26944
dev = "cuda"
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, N, K in [("qkv", 2304, 768), ("out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]:
    a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
    dy = torch.randn(M, N, device=dev).bfloat16()
    us = t(lambda: a @ w.t())
    print(f"blas nt {name}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TFLOP/s")
    us = t(lambda: dy.t() @ a)
    print(f"blas tn {name}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TFLOP/s")
