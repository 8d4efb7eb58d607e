"""Persistent 256x256 gemm_nt against an fp32 matmul of the same 16-bit operands; timing of the encoder shapes with it on/off
(SIMVG_GEMM_PERSIST=0 in a second process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops

dev = "cuda"
torch.manual_seed(0)
worst = 0.0
for (M, SPLIT, N, K, use_bias) in [(26944, 25664, 3072, 768, True), (26944, 25664, 2304, 768, True), (26944, 25664, 3072, 768, False),
                                    (26944, 0, 2304, 768, True), (4100, 3000, 2304, 128, True), (13472, 12832, 4096, 1024, True)]:
    a = torch.randn(M, K, device=dev).to(ops.LP())
    w = (torch.randn(2, N, K, device=dev) * K ** -0.5).to(ops.LP())
    bias = torch.randn(2, N, device=dev) if use_bias else None
    out = torch.full((M, N), float("nan"), device=dev, dtype=ops.LP())
    for rep in range(3):
        ops.gemm_nt(a, w, bias=bias, out=out, split=SPLIT)
    torch.cuda.synchronize()
    sp = SPLIT if SPLIT else M
    ref = torch.empty(M, N, device=dev)
    ref[:sp] = a[:sp].float() @ w[0].float().T + (bias[0] if use_bias else 0)
    if sp < M:
        ref[sp:] = a[sp:].float() @ w[1].float().T + (bias[1] if use_bias else 0)
    err = (out.float() - ref).abs().max().item()
    rnd = (ref.to(ops.LP()).float() - ref).abs().max().item()
    worst = max(worst, err)
    print(f"M={M} split={SPLIT} N={N} K={K} bias={use_bias}: max err {err:.3e} (rounding alone {rnd:.3e}) nan={torch.isnan(out.float()).any().item()}", flush=True)
    assert err <= 2.5 * rnd + 1e-6, "mismatch"
print("ok", worst)
