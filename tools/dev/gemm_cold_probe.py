"""gemm_nt back-to-back with operands that are NOT the previous launch's: rotating over 12 weight sets (the 12 layers) and / or 12
activation buffers -- which of them makes a standalone loop lose what the in-situ launches lose (tools/dev/gemm_insitu_table.py)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops
M, SPLIT, dev = 26944, 25664, "cuda"
SHAPES = [("qkv forward", 2304, 768, False), ("out-proj forward (fp32 + residual)", 768, 768, True), ("dgrad out-proj", 768, 768, False),
          ("dgrad qkv", 768, 2304, False), ("fc2 forward (fp32 + residual)", 768, 3072, True)]
for what, N, K, f32res in SHAPES:
    row = []
    for nw, na in ((1, 1), (12, 1), (1, 3), (12, 3)):
        a = [torch.randn(M, K, device=dev).to(ops.LP()) for _ in range(na)]
        w = [(torch.randn(2, N, K, device=dev) * K ** -0.5).to(ops.LP()) for _ in range(nw)]
        bias = torch.randn(2, N, device=dev)
        out = [torch.empty(M, N, device=dev, dtype=torch.float32 if f32res else ops.LP()) for _ in range(na)]
        resid = [torch.randn(M, N, device=dev) if f32res else None for _ in range(na)]

        def loop(reps):
            for i in range(reps):
                ops.gemm_nt(a[i % na], w[i % nw], bias=bias, out=out[i % na], split=SPLIT, residual=resid[i % na])
        loop(24)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); loop(240); e1.record(); e1.synchronize()
        row.append(e0.elapsed_time(e1) / 240 * 1e3)
        del a, w, out, resid
    print(f"{what:38s} same W, same A {row[0]:6.1f} | 12 W {row[1]:6.1f} | 3 A / out {row[2]:6.1f} | 12 W, 3 A / out {row[3]:6.1f} us", flush=True)
