"""The decoder's cross-attention on few queries (csrc/head.hip: attn_small_fwd / _bwd), 400 image keys, B = 64, 8 heads x 32:
python tools/dev/attn_small_bench.py [num_queries]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops
B, H, Lk, Nv = 64, 8, 400, 401
Lq = int(sys.argv[1]) if len(sys.argv) > 1 else 1
E = H * 32
dev = "cuda"
q = torch.randn(B * Lq, E, device=dev)
kv = torch.randn(B * Nv, 2 * E, device=dev)          # projected memory: K | V columns, CLS row carried along (kv_rows = Nv)
k, v = kv[:, :E], kv[:, E:]
kpos = torch.randn(Lk, E, device=dev)
kpm = torch.zeros(B, Lk, device=dev, dtype=torch.uint8)


def timed(name, fn, nbytes, reps=200):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:28s} {us:7.1f} us   {nbytes / us / 1e6:5.2f} TB/s ({nbytes / 1e6:.0f} MB)")


out, P = ops.attn_small_fwd(q, k[1:], v[1:], B, H, Lq, Lk, kpm=kpm, kv_rows=Nv, kpos=kpos)
timed("attn_small_fwd", lambda: ops.attn_small_fwd(q, k[1:], v[1:], B, H, Lq, Lk, kpm=kpm, kv_rows=Nv, kpos=kpos), B * Lk * E * 8.0)
dout = torch.randn_like(out)
dq = torch.zeros_like(q)
dkv = torch.zeros_like(kv)
timed("attn_small_bwd", lambda: ops.attn_small_bwd(q, k[1:], v[1:], P, dout, dq, dkv[1:, :E], dkv[1:, E:], B, H, Lq, Lk, kpm=kpm, kv_rows=Nv, kpos=kpos),
      B * Lk * E * 16.0)
