"""The head's fp32 GEMM shapes at B = 64, num_queries = 1 (csrc/head.hip): python tools/dev/head_gemm_bench.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = "cuda"
E, F = 256, 2048


def timed(name, fn, reps=300):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    print(f"{name:52s} {e0.elapsed_time(e1) / reps * 1e3:7.1f} us")


x, h1 = torch.randn(M, E, device=dev), torch.randn(M, F, device=dev)
W1, W2 = torch.randn(F, E, device=dev), torch.randn(E, F, device=dev)
b1, b2 = torch.randn(F, device=dev), torch.randn(E, device=dev)
o1, o2 = torch.empty(M, F, device=dev), torch.empty(M, E, device=dev)
m1, m2 = torch.ones(M, F, device=dev), torch.ones(M, E, device=dev)
timed("linear1 fwd  [M,256]x[256,2048] relu", lambda: ops.gemm_f32(x, E, 1, W1, 1, E, o1, M, F, E, bias=b1, act=2))
timed("linear2 fwd  [M,2048]x[2048,256] + addend", lambda: ops.gemm_f32(h1, F, 1, W2, 1, F, o2, M, E, F, bias=b2, addend=x, addend_rows=M))
timed("linear2 fwd as a group of one (mult, addend)", lambda: ops.gemm_f32_group([ops.gp(h1, F, 1, W2, 1, F, o2, M, E, F, bias=b2, mult=m2, addend=x, addend_rows=M)]))
# backward of linear2: dgrad [M,256]x[256,2048] (B = W2 read with k along its rows), wgrad [256,M]x[M,2048]
dW2, dh = torch.zeros(E, F, device=dev), torch.empty(M, F, device=dev)
timed("linear2 dgrad [M,256]x[256,2048]", lambda: ops.gemm_f32(o2, E, 1, W2, F, 1, dh, M, F, E))
timed("linear2 wgrad [256,M]x[M,2048]", lambda: ops.gemm_f32(o2, 1, E, h1, F, 1, dW2, E, F, M))
