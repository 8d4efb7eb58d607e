"""Phase timeline of dec_attn_fwd_kernel (workgroup 0, 100 MHz wall clock), development build:
    SIMVG_EXTRA_FLAGS=-DDEC_TIMELINE SIMVG_LIB_SUFFIX=_tl python -m simvg_amd.build
    SIMVG_HIP_LIB=simvg_amd/lib/libsimvg_hip_tl.so python tools/dev/decoder_timeline.py [R]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from simvg_amd import hip_ops as ops, _lib
    import test_decoder_gpu as T
    lib = ctypes.CDLL(_lib.LIB_PATH)
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    E = 256
    M = B * R
    for kind, Lk in (("mem", 400), ("text", 20)):
        W = T._layer_params(g)
        Wl = [W[k].to(dev) for k in T.ATTN_KEYS]
        tgt, qpos = torch.randn(M, E, generator=g).to(dev), torch.randn(M, E, generator=g).to(dev)
        kv_rows, kv_off = (Lk + 1, 1) if kind == "mem" else (Lk, 0)
        src = torch.randn(B * kv_rows, E, generator=g).to(dev)
        if kind == "mem":
            src = src.to(ops.LP())
        kpos = torch.randn(Lk, E, generator=g).to(dev)
        for _ in range(5):
            ops.dec_attn_fwd(tgt, qpos, Wl, src, B, R, Lk, kv_rows=kv_rows, kv_off=kv_off, kpos=kpos)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 64)()
        assert lib.simvg_dec_timeline(buf) == 0
        t = list(buf)
        marks = [(i, v) for i, v in enumerate(t) if v]
        print(f"== {kind} Lk={Lk} R={R}: total {(max(v for _, v in marks) - t[0]) / 100:.2f} us")
        prev = t[0]
        for i, v in sorted(marks, key=lambda iv: iv[1]):
            print(f"   mark {i:2d}: +{(v - prev) / 100:6.2f} us   (at {(v - t[0]) / 100:7.2f})")
            prev = v


if __name__ == "__main__":
    main()
