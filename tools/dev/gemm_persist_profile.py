"""Per-tile timeline of the persistent gemm_nt (library built with -DSIMVG_PQ_PROFILE): s_memtime of wave 0 at
0 tile top, 1 after the first wait, 2/3 around the k-tile-2 wait (stores drained), 4 end of the k loop, 5 after the extra
barrier + issue, 6 end of the epilogue.  Ticks are s_memtime units (calibrated against the event time of the launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
prof = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
os.environ["SIMVG_PQ_PROF_PTR"] = str(prof.data_ptr())
from simvg_amd import hip_ops as ops
M, SPLIT = 26944, 25664
name, N, K = {"fc1": ("fc1", 3072, 768), "qkv": ("qkv", 2304, 768)}[sys.argv[1] if len(sys.argv) > 1 else "fc1"]
a = torch.randn(M, K, device="cuda").to(ops.LP())
w = (torch.randn(2, N, K, device="cuda") * K ** -0.5).to(ops.LP())
bias = torch.randn(2, N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=ops.LP())
for _ in range(20):
    ops.gemm_nt(a, w, bias=bias, out=out, split=SPLIT)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.gemm_nt(a, w, bias=bias, out=out, split=SPLIT)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
p = prof.cpu().view(256, 8, 8).double()      # ticks
last = (p[:, :, 6] > 0).sum(1) - 1
span = torch.stack([p[i, last[i], 6] - p[i, 0, 0] for i in range(256)])
print(f"kernel {us:.1f} us by events; WG span {span.mean():.0f} ticks (max {span.max():.0f}) -> {span.max() / us / 1e3:.3f} ticks per ticks")
t0 = p[:, 0, 0].min()
for wg in (0, 1, 8, 100, 255):
    print(f"WG {wg}:")
    for t in range(6):
        if p[wg, t, 6] == 0:
            continue
        r = p[wg, t] - t0
        print(f"  tile {t}: top {r[0]:9.0f} | wait0 {r[1]-r[0]:6.0f} | k0-k1 {r[2]-r[1]:6.0f} | wait2 {r[3]-r[2]:6.0f} | k2..end {r[4]-r[3]:6.0f} | bar+issue {r[5]-r[4]:6.0f} | epilogue {r[6]-r[5]:6.0f} ticks")
valid = p[:, :, 6] > 0
d = lambda i, j: (p[:, :, i] - p[:, :, j])[valid]
for nm, x in (("wait0", d(1, 0)), ("k0-k1", d(2, 1)), ("wait2", d(3, 2)), ("k2..end", d(4, 3)), ("bar+issue", d(5, 4)), ("epilogue", d(6, 5))):
    print(f"{nm:10s} mean {x.mean():7.0f} ticks  p50 {x.median():7.0f}  max {x.max():7.0f}")
print("tiles per WG:", valid.sum(1).float().mean().item())
