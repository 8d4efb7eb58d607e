"""A/B of gemm_nt variants on the N = 768 launches of a training step (GPU box): the forward ones write the fp32 residual
stream (residual epilogue, K = 768 / 3072), the dgrads write 16-bit (K = 768 / 2304 / 3072).  Variants are picked by the
launcher from SIMVG_GEMM_N768 on every call, so they interleave inside one process (rounds x variants, min and median)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops

M, SPLIT, N = 26944, 25664, 768
VARIANTS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["base", "respf0"]
REPS, ROUNDS = 200, 3
ROT = int(os.environ.get("ROT", "1"))      # > 1: rotate that many activation / residual / output sets (a back-to-back loop over ONE
                                           # set keeps them in the 256 MB Infinity Cache: profiles/r05_sweeps.md section 11)
dev = "cuda"
cases = [("out-proj fwd  K=768  f32+res", 768, True), ("fc2 fwd       K=3072 f32+res", 3072, True),
         ("dgrad out     K=768  16-bit", 768, False), ("dgrad qkv     K=2304 16-bit", 2304, False), ("dgrad fc1     K=3072 16-bit", 3072, False)]
tot = {v: 0.0 for v in VARIANTS}
for name, K, res in cases:
    a = torch.randn(M, K, device=dev).to(ops.LP())
    w = (torch.randn(2, N, K, device=dev) * K ** -0.5).to(ops.LP())
    bias = torch.randn(2, N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if res else ops.LP())
    sets = [(a, r, out)] + [(a.clone(), None if r is None else r.clone(), torch.empty_like(out)) for _ in range(ROT - 1)]
    ref = None
    t = {v: [] for v in VARIANTS}
    for rnd in range(ROUNDS):
        for v in VARIANTS:
            for kk in ("SIMVG_GEMM_N768", "SIMVG_GEMM_RESPF", "SIMVG_GEMM_PP", "SIMVG_GEMM_320", "SIMVG_GEMM_TALL"):
                os.environ.pop(kk, None)
            for kv in ([] if v == "base" else v.split("+")):      # "respf0"  (the 8-wave variant "w8" of r04_sweeps.md section 2 is
                if kv == "respf0":                                # no longer built: tools/dev/gemm_variants_r04.hip.txt)
                    os.environ["SIMVG_GEMM_RESPF"] = "0"
                if kv == "notall":                                # round 6: without the one-round 320-row kernel (gemm_nt_kernel_tall)
                    os.environ["SIMVG_GEMM_TALL"] = "0"
                if kv == "t320":                                  # one round of 320 x 256 tiles (round 4; not built: tools/dev/gemm_320x256_r04.hip.txt)
                    os.environ["SIMVG_GEMM_320"] = "1"
                if kv == "pp":                                    # the ping-pong form of the 160 x 256 kernel (round 4; not built any
                    os.environ["SIMVG_GEMM_PP"] = "1"             # more: tools/dev/gemm_pingpong_r04.hip.txt)
            for _ in range(10):
                ops.gemm_nt(a, w, bias=bias, out=out, split=SPLIT, residual=r)
            torch.cuda.synchronize()
            if rnd == 0:
                if ref is None:
                    ref = out.float().clone()
                else:
                    err = float((out.float() - ref).abs().max()) / float(ref.abs().max())
                    assert err <= 2e-3, (name, v, err)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for it in range(REPS):
                a_, r_, o_ = sets[it % ROT]
                ops.gemm_nt(a_, w, bias=bias, out=o_, split=SPLIT, residual=r_)
            e1.record(); torch.cuda.synchronize()
            t[v].append(e0.elapsed_time(e1) / REPS * 1e3)
    line = f"{name}: "
    for v in VARIANTS:
        us = min(t[v]); tot[v] += us * (12 if True else 0)
        line += f" {v}: {us:6.1f} us ({2.0 * M * N * K / us / 1e6:6.0f} TF/s, median {sorted(t[v])[len(t[v]) // 2]:6.1f})"
    print(line, flush=True)
print("sum over the five shapes x 12 layers (ms/step): " + "  ".join(f"{v}: {tot[v] / 1e3:.3f}" for v in VARIANTS))
