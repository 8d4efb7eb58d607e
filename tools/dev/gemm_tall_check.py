"""GPU: the one-round 320-row kernel (gemm_nt_kernel_tall) against an fp32 matmul of the same 16-bit operands, every epilogue it
takes, ragged row groups, split weights; then timing of the five N = 768 launch forms with it on / off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops

dev = "cuda"
torch.manual_seed(0)
CASES = [("1", (26944, 25664, 768, 768)), ("1", (26944, 25664, 768, 128)), ("1", (20011, 18003, 768, 192)), ("1", (16000, 0, 768, 64)),
         ("1", (60000, 58000, 256, 128)),
         # the 2 x 8-wave 224-row body (gemm_nt_kernel_t224_*): ViT-L's row count, ragged groups, one group
         (None, (13472, 12832, 1024, 256)), (None, (10300, 9800, 1024, 128)), (None, (9900, 0, 1024, 64)), (None, (13472, 12832, 1024, 1024))]
for (TALL, (M, SPLIT, N, K)) in CASES:
    if TALL is None:
        os.environ.pop("SIMVG_GEMM_TALL", None)
    else:
        os.environ["SIMVG_GEMM_TALL"] = TALL
    a = torch.randn(M, K, device=dev).to(ops.LP())
    w = (torch.randn(2, N, K, device=dev) * K ** -0.5).to(ops.LP())
    bias = torch.randn(2, N, device=dev)
    sp = SPLIT if SPLIT else M
    ref = torch.empty(M, N, device=dev)
    ref[:sp] = a[:sp].float() @ w[0].float().T + bias[0]
    if sp < M:
        ref[sp:] = a[sp:].float() @ w[1].float().T + bias[1]
    out = torch.full((M, N), float("nan"), device=dev, dtype=ops.LP())
    ops.gemm_nt(a, w, bias=bias, out=out, split=SPLIT)
    torch.cuda.synchronize()
    err = (out.float() - ref).abs().max().item()
    rnd = (ref.to(ops.LP()).float() - ref).abs().max().item()
    print(f"16-bit  M={M} split={SPLIT} N={N} K={K}: max err {err:.3e} (rounding alone {rnd:.3e})", flush=True)
    assert err <= 2.5 * rnd + 1e-6 and not torch.isnan(out.float()).any()
    res = torch.randn(M, N, device=dev)
    rps = (401, 20) if SPLIT else (M, 1)
    ns = max((sp + rps[0] - 1) // rps[0], (M - sp + rps[1] - 1) // rps[1] if SPLIT else 1)
    scale = torch.rand(ns, device=dev) + 0.5
    rows = torch.arange(M, device=dev)
    samp = torch.where(rows < sp, rows // rps[0], (rows - sp) // rps[1])
    o32 = torch.full((M, N), float("nan"), device=dev)
    ops.gemm_nt(a, w, bias=bias, out=o32, split=SPLIT, residual=res, row_scale=scale, rows_per_sample=rps)
    e1 = (o32 - (res + scale[samp][:, None] * ref)).abs().max().item()
    o32.fill_(float("nan"))
    ops.gemm_nt(a, w, bias=bias, out=o32, split=SPLIT, residual=res)
    e2 = (o32 - (res + ref)).abs().max().item()
    o32.fill_(float("nan"))
    ops.gemm_nt(a, w, bias=bias, out=o32, split=SPLIT)
    e3 = (o32 - ref).abs().max().item()
    print(f"fp32    residual*scale {e1:.3e}  residual {e2:.3e}  plain {e3:.3e}", flush=True)
    assert max(e1, e2, e3) <= 2e-4, (e1, e2, e3)
    # bit-identical to the other kernels' results (same MFMA order per accumulator along k)
    os.environ["SIMVG_GEMM_TALL"] = "0"
    os.environ["SIMVG_GEMM_T224"] = "0"
    os.environ["SIMVG_GEMM_TALL4"] = "0"
    o2 = torch.empty_like(o32)
    ops.gemm_nt(a, w, bias=bias, out=o2, split=SPLIT)
    for kk in ("SIMVG_GEMM_TALL", "SIMVG_GEMM_T224", "SIMVG_GEMM_TALL4"):
        os.environ.pop(kk, None)
    if TALL is not None:
        os.environ["SIMVG_GEMM_TALL"] = TALL
    print("        vs the other kernels: max diff", (o2 - o32).abs().max().item(), flush=True)
# split weights (hi + lo)
os.environ["SIMVG_GEMM_TALL"] = "1"
M, SPLIT, N, K = 26944, 25664, 768, 768
a = torch.randn(M, K, device=dev).to(ops.LP())
wf = torch.randn(2, N, K, device=dev) * K ** -0.5
w2 = ops.split_weight(wf)
bias = torch.randn(2, N, device=dev)
res = torch.randn(M, N, device=dev)
o = torch.empty(M, N, device=dev)
ops.gemm_nt_split(a, w2, bias=bias, out=o, split=SPLIT, residual=res)
ref = torch.cat([a[:SPLIT].float() @ wf[0].T + bias[0], a[SPLIT:].float() @ wf[1].T + bias[1]], 0) + res
print("split weights fp32+res: max err", (o - ref).abs().max().item(), flush=True)
assert (o - ref).abs().max().item() <= 1e-4
print("ok")
