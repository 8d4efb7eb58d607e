"""per-head cost of the streamed attention forward: launch time at 3 and 6 heads per workgroup, with ablations"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
os.environ.setdefault("SIMVG_ATTN_STREAM", "1")
from simvg_amd import hip_ops as ops
dev = torch.device("cuda", 0)
H, N = 12, 421
res = {}
for B in (64, 128):
    qkv = (torch.randn(B * N, 3 * H * 64) * 0.5).to(dev).to(ops.LP())
    pad = torch.zeros(B, 20, dtype=torch.uint8); pad[:, 9:] = 1; pad = pad.to(dev)
    out, lse = ops.attn_fwd(qkv, B, H, 401, 20, pad=pad)
    for abl in [int(x) for x in os.environ.get("ABLS", "0,64,358").split(",")]:
        os.environ["SIMVG_STREAM_ABL"] = str(abl)
        for _ in range(5):
            ops.attn_fwd(qkv, B, H, 401, 20, pad=pad, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.attn_fwd(qkv, B, H, 401, 20, pad=pad, out=out)
        e1.record(); e1.synchronize()
        res[(B, abl)] = e0.elapsed_time(e1) * 20
for abl in sorted({k[1] for k in res}):
    a, b = res[(64, abl)], res[(128, abl)]
    print(f"abl {abl:4d}: 3 heads/WG {a:.1f} us, 6 heads/WG {b:.1f} us -> {(b - a) / 3:.2f} us per head, fixed {a - (b - a):.1f} us")
