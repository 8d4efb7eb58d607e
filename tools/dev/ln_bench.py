"""LayerNorm / Adam micro-benchmark: the hot instances of the encoder step at B=64 (M = 26944 rows, split 25664)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops

M, SPLIT, D, F = 26944, 25664, 768, 3072
dev = "cuda"
LP = ops.LP()
g = torch.Generator(device="cpu").manual_seed(0)


def r(*s, dt=torch.float32, sc=1.0):
    return (torch.randn(*s, generator=g) * sc).to(dev).to(dt)


def timed(name, fn, nbytes, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:44s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s (algorithmic {nbytes / 1e6:.0f} MB)", flush=True)


gam, bet = 1 + 0.1 * r(2, D), 0.1 * r(2, D)
gamF, betF = 1 + 0.1 * r(2, F), 0.1 * r(2, F)
x32, xlp, u = r(M, D), r(M, D, dt=LP), r(M, F, dt=LP)
h, g2 = torch.empty(M, D, device=dev, dtype=LP), torch.empty(M, F, device=dev, dtype=LP)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "fwd"):
    timed("ln_fwd fp32 residual -> lp (ln1/ln2)", lambda: ops.ln_fwd(x32, gam, bet, split=SPLIT, y=h), M * D * 6.0)
    timed("ln_fwd lp attn out -> lp (inner_attn_ln)", lambda: ops.ln_fwd(xlp, gam, bet, split=SPLIT, y=h), M * D * 4.0)
    timed("ln_fwd u -> lp, NO gelu (3072 wide)", lambda: ops.ln_fwd(u, gamF, betF, split=SPLIT, y=g2, gelu_in=False), M * F * 4.0)
    timed("ln_fwd gelu(u) -> lp (ffn_layernorm)", lambda: ops.ln_fwd(u, gamF, betF, split=SPLIT, y=g2, gelu_in=True), M * F * 4.0)
if which in ("all", "bwd"):
    _, _, m1, r1 = ops.ln_fwd(x32, gam, bet, split=SPLIT, y=h)
    _, _, m2, r2 = ops.ln_fwd(xlp, gam, bet, split=SPLIT, y=h)
    _, _, m4, r4 = ops.ln_fwd(u, gamF, betF, split=SPLIT, y=g2, gelu_in=True)
    dD, dF = r(M, D, dt=LP), r(M, F, dt=LP)
    dg, db = torch.zeros(2, D, device=dev), torch.zeros(2, D, device=dev)
    dgF, dbF = torch.zeros(2, F, device=dev), torch.zeros(2, F, device=dev)
    dx, dyb, dO, dF2 = r(M, D), torch.empty(M, D, device=dev, dtype=LP), torch.empty(M, D, device=dev, dtype=LP), torch.empty(M, F, device=dev, dtype=LP)
    timed("ln_bwd residual (ln1/ln2): dres+dx fp32, lp copy", lambda: ops.ln_bwd(dD, x32, m1, r1, gam, dg, db, split=SPLIT, dres=dx, dx_f32=dx, dx_scaled=dyb), M * D * (2 + 4 + 4 + 4 + 2.0))
    timed("ln_bwd inner_attn_ln: lp x, lp dx", lambda: ops.ln_bwd(dD, xlp, m2, r2, gam, dg, db, split=SPLIT, dx_lp=dO), M * D * 6.0)
    timed("ln_bwd ffn_layernorm (gelu recompute)", lambda: ops.ln_bwd(dF, u, m4, r4, gamF, dgF, dbF, split=SPLIT, dx_lp=dF2, gelu_u=u), M * F * 6.0)
if which in ("all", "adam"):
    n = 141_000_000 // 4 * 4
    p, gr, m, v, vm = (torch.randn(n, device=dev) for _ in range(5))
    v.abs_(); vm.abs_()
    tn = torch.ones(1, device=dev)
    timed("adam_step amsgrad 141 M (dense)", lambda: ops.adam_step(p, gr, m, v, vm, 1e-4, 0.9, 0.9, 0.98, 1e-9, total_norm=tn, max_norm=0.15), n * 36.0, reps=10)
    z = int(n * 0.65) // 4 * 4                       # ViT-B: the last 35 % of the arena is the (mostly untouched) text table
    for t in (gr, m, v, vm):
        t[z:].zero_()
    timed("adam_step amsgrad 141 M (35 % untouched rows)", lambda: ops.adam_step(p, gr, m, v, vm, 1e-4, 0.9, 0.9, 0.98, 1e-9, total_norm=tn, max_norm=0.15), z * 36.0 + (n - z) * 12.0, reps=10)
    sq = torch.zeros(1, device=dev)
    timed("sumsq 141 M", lambda: ops.sumsq_accum(gr, sq), n * 4.0, reps=10)
