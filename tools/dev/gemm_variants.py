import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops
M, SPLIT = 26944, 25664
N, K = int(os.environ.get("N", 768)), int(os.environ.get("K", 3072))
a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(2, N, K, device="cuda") * K ** -0.5).bfloat16()
bias = torch.randn(2, N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
outf = torch.empty(M, N, device="cuda", dtype=torch.float32)
res = torch.randn(M, N, device="cuda")
def run(name, fn):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:34s}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TFLOP/s", flush=True)
for rep in range(3):
  run("plain 1 group", lambda: ops.gemm_nt(a, w[:1], out=out))
  run("bias 1 group", lambda: ops.gemm_nt(a, w[:1], bias=bias[:1], out=out))
  run("split 2 groups", lambda: ops.gemm_nt(a, w, out=out, split=SPLIT))
  run("split+bias", lambda: ops.gemm_nt(a, w, bias=bias, out=out, split=SPLIT))
  run("split+bias+res f32 out", lambda: ops.gemm_nt(a, w, bias=bias, out=outf, split=SPLIT, residual=res))
  run("split+bias+gelu+aux", lambda: ops.gemm_nt(a, w, bias=bias, out=out, split=SPLIT, act=1, aux_preact=outf.view(torch.bfloat16)[:, :N]))
