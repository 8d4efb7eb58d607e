"""wgrad with and without the fused bias gradient, interleaved (GPU box): what the column sums cost the XCD-partitioned kernel"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops
dev = "cuda"
LP = ops.LP()
M, SPLIT = 64 * 421, 64 * 401
REPS = 300
for name, N, K in [("qkv", 2304, 768), ("fc1", 3072, 768), ("fc2", 768, 3072), ("out", 768, 768)]:
    dy = torch.randn(M, N, device=dev).to(LP)
    x = torch.randn(M, K, device=dev).to(LP)
    dw = torch.zeros(2, N, K, device=dev)
    db = torch.zeros(2, N, device=dev)
    res = {}
    for rnd in range(3):
        for tag, d in (("with db", db), ("no db", None)):
            for _ in range(10):
                ops.gemm_tn(dy, x, dw, split=SPLIT, db=d)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                ops.gemm_tn(dy, x, dw, split=SPLIT, db=d)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(tag, []).append(e0.elapsed_time(e1) / REPS * 1e3)
    print(f"wgrad {name:4s} [{M}x{N}x{K}] " + "  ".join(f"{t}: {min(v):.1f} us (median {sorted(v)[1]:.1f})" for t, v in res.items()), flush=True)
