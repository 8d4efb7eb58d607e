"""Is an HBM-bound kernel slower right after a burst of MFMA-bound work?  Adam (141 M parameters) timed back to back, then
timed immediately after 40 fc1-shaped GEMM launches (~5 ms of MFMA work), as it runs in the training step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops

dev = "cuda"
n = 141_000_000 // 4 * 4
p, gr, m, v, vm = (torch.randn(n, device=dev) for _ in range(5))
v.abs_(); vm.abs_()
tn = torch.ones(1, device=dev)
M, SPLIT, N, K = 26944, 25664, 3072, 768
a = torch.randn(M, K, device=dev).to(ops.LP())
w = (torch.randn(2, N, K, device=dev) * K ** -0.5).to(ops.LP())
out = torch.empty(M, N, device=dev, dtype=ops.LP())
x32 = torch.randn(M, 768, device=dev)


def adam():
    ops.adam_step(p, gr, m, v, vm, 1e-4, 0.9, 0.9, 0.98, 1e-9, total_norm=tn, max_norm=0.15)


def timed(fn, pre=None, reps=10):
    ts = []
    for _ in range(reps):
        if pre:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def burst(k):
    def f():
        for _ in range(k):
            ops.gemm_nt(a, w, out=out, split=SPLIT)
    return f


def copy():
    x32.mul_(1.0001)


for _ in range(3):
    adam()
print(f"adam alone (one launch at a time, idle GPU before): {timed(adam):8.1f} us")
print(f"adam after 40 GEMMs (~5 ms MFMA):                   {timed(adam, burst(40)):8.1f} us")
print(f"adam after 200 GEMMs (~25 ms MFMA):                 {timed(adam, burst(200)):8.1f} us")
print(f"gemm fc1 alone:                                     {timed(burst(1)):8.1f} us")
print(f"gemm fc1 after 200 GEMMs:                           {timed(burst(1), burst(200)):8.1f} us")
print(f"fp32 scale of 83 MB alone:                          {timed(copy):8.1f} us")
print(f"fp32 scale of 83 MB after 200 GEMMs:                {timed(copy, burst(200)):8.1f} us")
# sustained load: ~2 s of GEMMs, then Adam / the GEMM itself, three times
for _ in range(3):
    burst(16000)()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(); adam(); e1.record(); burst(1)(); e2.record(); e2.synchronize()
    print(f"after ~2 s of sustained GEMMs: adam {e0.elapsed_time(e1) * 1e3:8.1f} us, gemm fc1 {e1.elapsed_time(e2) * 1e3:8.1f} us")
# Adam with the training step's state: sparse gradient rows
z = int(n * 0.65) // 4 * 4
for t in (gr, m, v, vm):
    t[z:].zero_()
print(f"adam, last 35 % untouched, alone:                  {timed(adam):8.1f} us")
# gradients with many exact zeros inside touched rows (e.g. masked tokens): does the per-float4 test thrash?
gr[:z].mul_((torch.rand(z, device=dev) > 0.5).float())
print(f"adam, + half of the touched gradients zero:        {timed(adam):8.1f} us")
