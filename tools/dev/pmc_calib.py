"""Known-byte launches to calibrate FETCH_SIZE / WRITE_SIZE for this build's access patterns (run under
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, see tools/dev/pmc_calib.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops
dev = "cuda"
M = 26944
# (a) read-only stream, 16-B global loads: sumsq over 564 MB
g = torch.randn(141_000_000, device=dev)
sq = torch.zeros(1, device=dev)
# (b) gemm_nt with ONE column tile: A [M, 768] bf16 (41.4 MB) is read exactly once, W 0.4 MB, out [M, 256] bf16 13.8 MB
a = torch.randn(M, 768, device=dev).to(torch.bfloat16)
w = torch.randn(256, 768, device=dev).to(torch.bfloat16)
out = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
# (c) gemm_tn with one output tile column block: dW[256, 128] += dY[M, 256]^T X[M, 128]: dY 13.8 MB + X 6.9 MB read once
dy = torch.randn(M, 256, device=dev).to(torch.bfloat16)
x = torch.randn(M, 128, device=dev).to(torch.bfloat16)
dw = torch.zeros(256, 128, device=dev)
big = torch.empty(300_000_000, device=dev)          # 1.2 GB: flushes the 256 MB Infinity Cache between launches
for _ in range(3):
    big.fill_(1.0)
    ops.sumsq_accum(g, sq)
    big.fill_(2.0)
    ops.gemm_nt(a, w, out=out)
    big.fill_(3.0)
    ops.gemm_tn(dy, x, dw)
torch.cuda.synchronize()
print("known bytes: sumsq read 564.0 MB | gemm_nt read 41.4 + 0.4 MB, write 13.8 MB | gemm_tn read 13.8 + 6.9 MB")
