"""The fused decoder-layer kernels (csrc/decoder.hip) alone, at the bench geometry: run under rocprofv3 for per-kernel times.
    cd /tmp && rocprofv3 --kernel-trace --stats -d out -- python tools/dev/decoder_bench.py [B] [R]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from simvg_amd import hip_ops as ops
    import test_decoder_gpu as T
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    E, H = 256, 8
    M = B * R
    for kind, Lk, ffn in (("mem", 400, 2048), ("text", 20, 512)):
        W = T._layer_params(g, ffn=ffn)
        Wl = [W[k].to(dev) for k in T.ATTN_KEYS]
        tgt, qpos = torch.randn(M, E, generator=g).to(dev), torch.randn(M, E, generator=g).to(dev)
        kv_rows, kv_off = (Lk + 1, 1) if kind == "mem" else (Lk, 0)
        src = torch.randn(B * kv_rows, E, generator=g).to(dev)
        if kind == "mem":
            src = src.to(ops.LP())
        kpos = torch.randn(Lk, E, generator=g).to(dev)
        dt2 = torch.randn(M, E, generator=g).to(dev)
        dsrc = torch.empty(B * kv_rows, E, device=dev)
        kw = dict(kv_rows=kv_rows, kv_off=kv_off, kpos=kpos)
        W1, b1, W2, b2, g2, b2n = (W[k].to(dev) for k in ("W1", "b1f", "W2", "b2f", "g2", "b2"))
        for _ in range(20):
            sa = ops.dec_attn_fwd(tgt, qpos, Wl, src, B, R, Lk, **kw)
            sf = ops.dec_ffn_fwd(sa["t2"], W1, b1, W2, b2, g2, b2n, gP=g2, bP=b2n)
            d_r3, slabs, gf = ops.dec_ffn_bwd(sf, sa["t2"], W1, W2, g2, gP=g2, d_t3=dt2, d_hs=dt2)
            ops.dec_attn_bwd(sa, tgt, qpos, Wl, src, B, R, Lk, dt2=d_r3, dt2_slabs=slabs, dsrc=dsrc, **kw)
        torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
