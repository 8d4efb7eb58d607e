"""Encoder attention micro-benchmark at the bench geometry (B=64, H=12, 401+20 tokens)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops
B, H, Nv, T, D = 64, 12, 401, 20, 768
M = B * (Nv + T)
qkv = (torch.randn(M, 3 * D, device="cuda") * 0.5).bfloat16()
pad = torch.zeros(B, T, dtype=torch.uint8, device="cuda"); pad[:, 12:] = 1
out = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
dout = (torch.randn(M, D, device="cuda") * 0.1).bfloat16()
dqkv = torch.empty(M, 3 * D, device="cuda", dtype=torch.bfloat16)
def t(fn, n=20):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
_, lse = ops.attn_fwd(qkv, B, H, Nv, T, pad=pad, out=out)
print(f"attn fwd {t(lambda: ops.attn_fwd(qkv, B, H, Nv, T, pad=pad, out=out)):.1f} us   bwd {t(lambda: ops.attn_bwd(qkv, out, dout, lse, B, H, Nv, T, pad=pad, dqkv=dqkv)):.1f} us")
