"""a handful of encoder-attention forward launches at the bench geometry (target of rocprofv3 --pmc passes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops
dev = torch.device("cuda", 0)
B, H, N = int(os.environ.get("B", 64)), int(os.environ.get("H", 12)), 421
qkv = (torch.randn(B * N, 3 * H * 64) * 0.5).to(dev).to(ops.LP())
pad = torch.zeros(B, 20, dtype=torch.uint8); pad[:, 9:] = 1; pad = pad.to(dev)
for _ in range(int(os.environ.get("REPS", 4))):
    out, lse = ops.attn_fwd(qkv, B, H, 401, 20, pad=pad)
torch.cuda.synchronize()
