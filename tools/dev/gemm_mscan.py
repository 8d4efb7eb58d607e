"""fc2-shaped NT GEMM (N=768, K=3072) at several M: how much does tile-count quantisation cost?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops
N, K = int(os.environ.get("N", 768)), int(os.environ.get("K", 3072))
for M in [26944, 21760, 32768, 43520, 65536]:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(1, N, K, device="cuda") * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(2): ops.gemm_nt(a, w, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.gemm_nt(a, w, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    tiles = -(-M // 256) * -(-N // 128)
    print(f"M={M} tiles={tiles} rounds={tiles/256:.2f}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TFLOP/s", flush=True)
