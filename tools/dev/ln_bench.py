import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops
M, SPLIT = 26944, 25664
dev = "cuda"
for D, xdt in [(3072, torch.bfloat16), (768, torch.float32), (768, torch.bfloat16)]:
    x = torch.randn(M, D, device=dev).to(xdt)
    g, b = torch.ones(2, D, device=dev), torch.zeros(2, D, device=dev)
    y, _, mean, rstd = ops.ln_fwd(x, g, b, split=SPLIT)
    dy = torch.randn(M, D, device=dev).to(torch.bfloat16)
    u = torch.randn(M, D, device=dev).to(torch.bfloat16)
    dg, db = torch.zeros(2, D, device=dev), torch.zeros(2, D, device=dev)
    dxb = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    dxf = torch.empty(M, D, device=dev)
    dres = torch.randn(M, D, device=dev)
    def run():
        if D == 3072:
            ops.ln_bwd(dy, u, mean, rstd, g, dg, db, split=SPLIT, dx_lp=dxb, gelu_u=None if os.environ.get('NOGELU') else u)   # x == u: GELU recompute path
        elif xdt == torch.float32:
            ops.ln_bwd(dy, x, mean, rstd, g, dg, db, split=SPLIT, dres=dres, dx_f32=dxf, dx_scaled=dxb)
        else:
            ops.ln_bwd(dy, x, mean, rstd, g, dg, db, split=SPLIT, dx_lp=dxb)
    for _ in range(30): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(f"ln_bwd D={D} x={xdt}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)

# forward of ffn_layernorm: LayerNorm(gelu(u)), bf16 in / bf16 out
u = torch.randn(M, 3072, device=dev).to(torch.bfloat16)
g, b = torch.ones(2, 3072, device=dev), torch.zeros(2, 3072, device=dev)
for _ in range(30): ops.ln_fwd(u, g, b, split=SPLIT, out_lp=True, out_f32=False, gelu_in=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.ln_fwd(u, g, b, split=SPLIT, out_lp=True, out_f32=False, gelu_in=True)
e1.record(); torch.cuda.synchronize()
print(f"ln_fwd(gelu) D=3072 bf16: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
