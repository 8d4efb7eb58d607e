"""The mid-size exact-fp32 GEMM problems of num_queries = 10 one by one (group of one, with the split-K workspace):
python tools/dev/head_gemm_mid.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops
dev = "cuda"


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def prob(M, N, K, o):
    A = torch.randn(M, K, device=dev) if o[0] == "K" else torch.randn(K, M, device=dev)
    B = torch.randn(N, K, device=dev) if o[1] == "K" else torch.randn(K, N, device=dev)
    Cm = torch.zeros(M, N, device=dev)
    sam, sak = (K, 1) if o[0] == "K" else (1, M)
    sbk, sbn = (1, K) if o[1] == "K" else (N, 1)
    return ops.gp(A, sam, sak, B, sbk, sbn, Cm, M, N, K)


shapes = [(640, 256, 256, "KK"), (640, 256, 256, "KN"), (256, 256, 640, "NN"), (1, 256, 640, "KN"), (640, 2048, 256, "KK"), (640, 2048, 256, "KN"),
          (256, 2048, 640, "NN"), (640, 256, 2048, "KK"), (640, 256, 2048, "KN"), (2048, 256, 640, "NN"), (1, 2048, 640, "KN"), (640, 512, 256, "KK"),
          (640, 256, 512, "KN"), (512, 256, 640, "NN"), (1280, 256, 256, "KK"), (1280, 256, 256, "KN"), (256, 256, 1280, "NN"), (1920, 256, 256, "KN"),
          (256, 256, 1920, "NN"), (1280, 256, 768, "KK"), (256, 768, 1280, "NN"), (1280, 768, 256, "KN")]
for M, N, K, o in shapes:
    q = prob(M, N, K, o)
    t = timed(lambda: ops.gemm_f32_group([q]))
    print(f"{M:5d}x{N:5d}x{K:5d} {o}  {t:7.1f} us   {2.0 * M * N * K / t / 1e6:7.2f} TFLOP/s", flush=True)
