"""A/B of wgrad_x variants (SIMVG_WGRAD read by the launcher on every call), interleaved rounds, the four encoder shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops
dev = "cuda"
LP = ops.LP()
B = int(os.environ.get("B", 64))
D = int(os.environ.get("D", 768))          # D=1024 B=32: the ViT-L shapes
M, SPLIT = B * 421, B * 401
REPS = 200
VARIANTS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["base", "s0"]
tot = {v: 0.0 for v in VARIANTS}
for name, N, K, per_layer in [("qkv", 3 * D, D, 1), ("fc1", 4 * D, D, 1), ("fc2", D, 4 * D, 1), ("out", D, D, 1)]:
    dy = torch.randn(M, N, device=dev).to(LP)
    x = torch.randn(M, K, device=dev).to(LP)
    db = torch.zeros(2, N, device=dev)
    res, ref = {}, None
    for rnd in range(3):
        for v in VARIANTS:
            for kk in ("SIMVG_WGRAD", "SIMVG_WG_PRIO", "SIMVG_WG_SLABS", "SIMVG_WGRAD_SQ", "SIMVG_WG_DBG", "SIMVG_WG_FV"):
                os.environ.pop(kk, None)
            for kv in ([] if v == "base" else v.split("+")):       # (the one-barrier "x1" and s_setprio "p*" variants of
                if kv == "s0":                                     #  r04_sweeps.md section 3: tools/dev/wgrad_variants_r04.hip.txt, not built)                                   # fp32 atomics instead of slabs + reduction launch
                    os.environ["SIMVG_WG_SLABS"] = "0"
                if kv == "sq":                                     # the 16-wave 256 x 256 kernel on every shape it divides
                    os.environ["SIMVG_WGRAD_SQ"] = "1"
                if kv.startswith("dbg"):                           # ablations of the 256 x 256 kernel: a build with -DSIMVG_WG_ABLATE
                                                                   # (SIMVG_EXTRA_FLAGS, SIMVG_HIP_LIB); wrong results: WG_NOCHECK=1
                    os.environ["SIMVG_WG_DBG"] = kv[3:]
                if kv.startswith("fv"):                            # virtual stages of the straddling partition's second flush
                    os.environ["SIMVG_WG_FV"] = kv[2:]
                if kv == "sq0":                                    # never (ViT-L shapes: the generic kernel of gemm.hip)
                    os.environ["SIMVG_WGRAD_SQ"] = "0"
            dw = torch.zeros(2, N, K, device=dev)
            db.zero_()
            ops.gemm_tn(dy, x, dw, split=SPLIT, db=db)
            torch.cuda.synchronize()
            if rnd == 0 and not os.environ.get("WG_NOCHECK"):
                if ref is None:
                    ref = (dw.clone(), db.clone())
                else:
                    e1 = float((dw - ref[0]).abs().max()) / float(ref[0].abs().max())
                    e2 = float((db - ref[1]).abs().max()) / float(ref[1].abs().max())
                    assert os.environ.get("WG_NOCHECK") or (e1 < 1e-5 and e2 < 1e-5), (name, v, e1, e2)
            for _ in range(10):
                ops.gemm_tn(dy, x, dw, split=SPLIT, db=db)
            torch.cuda.synchronize()
            e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                ops.gemm_tn(dy, x, dw, split=SPLIT, db=db)
            e1_.record(); torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1_) / REPS * 1e3)
    for v in VARIANTS:
        tot[v] += min(res[v]) * 12
    print(f"wgrad {name:4s} [{M}x{N}x{K}] " + "  ".join(f"{v}: {min(r):.1f} us ({2.0 * M * N * K / min(r) / 1e6:.0f} TF/s, median {sorted(r)[1]:.1f})" for v, r in res.items()), flush=True)
print("sum x 12 layers (ms/step): " + "  ".join(f"{v}: {tot[v] / 1e3:.3f}" for v in VARIANTS))
