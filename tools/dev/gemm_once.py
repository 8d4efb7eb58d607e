"""A handful of launches of each gemm_nt form of a training step (for PMC passes: tools/dev/pmc_kernels.sh python tools/dev/gemm_once.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops

M, SPLIT = 26944, 25664
dev = "cuda"
for name, N, K, res in [("qkv", 2304, 768, False), ("fc1", 3072, 768, False), ("dgrad_fc1", 768, 3072, False), ("dgrad_qkv", 768, 2304, False),
                        ("dgrad_out", 768, 768, False), ("fc2", 768, 3072, True), ("out", 768, 768, True)]:
    a = torch.randn(M, K, device=dev).to(ops.LP())
    w = (torch.randn(2, N, K, device=dev) * K ** -0.5).to(ops.LP())
    bias = torch.randn(2, N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if res else ops.LP())
    for _ in range(6):
        ops.gemm_nt(a, w, bias=bias, out=out, split=SPLIT, residual=r)
    torch.cuda.synchronize()
print("done")
