"""gemm_nt at inference sizes (forward_test B = 1 / 8): time per launch and error against fp32 matmul"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops

dev = "cuda"
shapes = [("qkv", 2304, 768), ("out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]
for B in (1, 2, 4, 8, 16, 32):
    M, SPLIT = B * 421, B * 401
    tot = 0.0
    for name, N, K in shapes:
        g = torch.Generator().manual_seed(N + K + B)
        a = torch.randn(M, K, generator=g).to(dev).to(ops.LP())
        w = (torch.randn(2, N, K, generator=g) * K ** -0.5).to(dev).to(ops.LP())
        bias = torch.randn(2, N, generator=g).to(dev)
        out = torch.empty(M, N, device=dev, dtype=ops.LP())
        ops.gemm_nt(a, w, bias=bias, out=out, split=SPLIT)
        ref = torch.cat([a[:SPLIT].float() @ w[0].float().t() + bias[0], a[SPLIT:].float() @ w[1].float().t() + bias[1]])
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        for _ in range(10):
            ops.gemm_nt(a, w, bias=bias, out=out, split=SPLIT)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.gemm_nt(a, w, bias=bias, out=out, split=SPLIT)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        tot += us
        print(f"B={B:2d} {name:4s} M={M:5d} N={N:4d} K={K:4d}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s  rel err {err:.1e}", flush=True)
    print(f"B={B:2d} sum of the four: {tot:.1f} us", flush=True)
