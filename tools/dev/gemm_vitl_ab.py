"""A/B of the 224-row gemm_nt tiles on the ViT-L launches of a training step at 32 pairs (M = 13 472): SIMVG_GEMM_224 = 0 (never) /
unset (the launcher's rule) / 1 (wherever N is a multiple of 256), switched per call inside one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops

B = int(os.environ.get("B", 32))
M, SPLIT = B * 421, B * 401
VARIANTS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["off", "rule"]
REPS, ROUNDS = 200, 3
dev = "cuda"
cases = [("out-proj fwd  N=1024 K=1024 f32+res", 1024, 1024, True, 1), ("fc2 fwd       N=1024 K=4096 f32+res", 1024, 4096, True, 1),
         ("dgrad out     N=1024 K=1024 16-bit", 1024, 1024, False, 1), ("dgrad qkv     N=1024 K=3072 16-bit", 1024, 3072, False, 1),
         ("dgrad fc1     N=1024 K=4096 16-bit", 1024, 4096, False, 1), ("qkv fwd       N=3072 K=1024 16-bit", 3072, 1024, False, 1),
         ("fc1 fwd / dgrad fc2 N=4096 K=1024 16-bit", 4096, 1024, False, 2)]
tot = {v: 0.0 for v in VARIANTS}
for name, N, K, res, per_layer in cases:
    a = torch.randn(M, K, device=dev).to(ops.LP())
    w = (torch.randn(2, N, K, device=dev) * K ** -0.5).to(ops.LP())
    bias = torch.randn(2, N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if res else ops.LP())
    ref = None
    t = {v: [] for v in VARIANTS}
    for rnd in range(ROUNDS):
        for v in VARIANTS:
            os.environ.pop("SIMVG_GEMM_224", None)
            os.environ.pop("SIMVG_GEMM_TALL4", None)
            os.environ.pop("SIMVG_GEMM_T224", None)
            if v == "not224":           # round 6: without the hand-managed 224-row kernel (2 x 8 waves)
                os.environ["SIMVG_GEMM_T224"] = "0"
            if v == "t224all":          # ... and with it wherever N is a multiple of 256, over several dispatch rounds too
                os.environ["SIMVG_GEMM_T224"] = "2"
            if v == "notall4":          # round 6: without the one-round 256-row hand-managed kernel for the fp32 epilogues
                os.environ["SIMVG_GEMM_TALL4"] = "0"
            if v == "off":
                os.environ["SIMVG_GEMM_224"] = "0"
            elif v == "on":
                os.environ["SIMVG_GEMM_224"] = "1"
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
            for _ in range(REPS):
                ops.gemm_nt(a, w, bias=bias, out=out, split=SPLIT, residual=r)
            e1.record(); torch.cuda.synchronize()
            t[v].append(e0.elapsed_time(e1) / REPS * 1e3)
    line = f"{name}: "
    for v in VARIANTS:
        us = min(t[v]); tot[v] += us * 24 * per_layer
        line += f" {v}: {us:6.1f} us ({2.0 * M * N * K / us / 1e6:6.0f} TF/s)"
    print(line, flush=True)
print("sum x 24 layers (ms/step): " + "  ".join(f"{v}: {tot[v] / 1e3:.3f}" for v in VARIANTS))
