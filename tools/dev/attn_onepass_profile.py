"""Phase timeline of the one-pass attention backward (development build: SIMVG_EXTRA_FLAGS="-DB1_PROFILE"
SIMVG_LIB_SUFFIX=_b1prof python -m simvg_amd.build; run with SIMVG_HIP_LIB=simvg_amd/lib/libsimvg_hip_b1prof.so
).  Workgroup 0 stamps s_memtime at 8 points of every query pair into the (otherwise unused) delta workspace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops, _lib
B, H, Nv, Nt, d = 64, 12, 401, 20, 64
N, D = Nv + Nt, H * d
dev = "cuda"
qkv = torch.randn(B * N, 3 * D, device=dev).to(ops.LP())
out, lse = ops.attn_fwd(qkv, B, H, Nv, Nt)
dout = torch.randn(B * N, D, device=dev).to(ops.LP()) * 0.01
dqkv = torch.empty_like(qkv)
delta = torch.zeros_like(lse)
lib = _lib.load()
p = ops._p
for _ in range(3):
    rc = lib.simvg_attn_bwd(p(qkv), qkv.stride(0), p(out), out.stride(0), p(dout), dout.stride(0), p(dqkv), dqkv.stride(0), p(lse),
                            p(delta), None, B, H, Nv, Nt, D, d ** -0.5, ops._stream())
torch.cuda.synchronize()
t = delta.view(torch.int32).cpu().view(-1)[:4 * 16 * 8].view(4, 16, 8).long() & 0xffffffff
x0 = int(t[:, 14, 0].min())
for w in range(4):
    r = [(int(v) - x0) & 0xffffffff for v in t[w, 14, :6]]
    print(f"wave {w}: entry {r[0]}, prologue done {r[1]}, barrier passed {r[2]}, pair loop done {r[3]}, last write-out {r[4]}, dK / dV stores issued {r[5]}")
names = ["frag reads + wait", "strip loop", "dQ phase", "wait barrier 1", "partial stores + commit", "wait barrier 2", "write-out"]
for w in range(4):
    seg = (t[w, 2:12, 1:] - t[w, 2:12, :-1]).float().mean(0)
    per = (t[w, 3:13, 0] - t[w, 2:12, 0]).float().mean()
    print(f"wave {w}: pair period {per:.0f} cycles; " + "; ".join(f"{n} {v:.0f}" for n, v in zip(names, seg)))
