"""A/B on the wide-N launches of a training step (qkv forward, fc1 forward / dgrad fc2; 16-bit output + bias): the persistent 256-row
kernel against gemm_nt_kernel_tall5 dispatched over several rounds (SIMVG_GEMM_TALL=2), rotating buffer sets (ROT)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops

M, SPLIT = 26944, 25664
ROT = int(os.environ.get("ROT", "3"))
REPS, ROUNDS = 150, 3
dev = "cuda"
for name, N, K in [("qkv fwd", 2304, 768), ("fc1 fwd / dgrad fc2", 3072, 768)]:
    sets = [((torch.randn(M, K, device=dev)).to(ops.LP()), torch.empty(M, N, device=dev, dtype=ops.LP())) for _ in range(ROT)]
    w = (torch.randn(2, N, K, device=dev) * K ** -0.5).to(ops.LP())
    bias = torch.randn(2, N, device=dev)
    t = {"persistent": [], "tall": []}
    ref = None
    for rnd in range(ROUNDS):
        for v in t:
            os.environ["SIMVG_GEMM_TALL"] = "2" if v == "tall" else "0"
            for _ in range(10):
                ops.gemm_nt(sets[0][0], w, bias=bias, out=sets[0][1], split=SPLIT)
            torch.cuda.synchronize()
            if rnd == 0:
                if ref is None:
                    ref = sets[0][1].float().clone()
                else:
                    print("   max diff vs persistent:", float((sets[0][1].float() - ref).abs().max()))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for it in range(REPS):
                a_, o_ = sets[it % ROT]
                ops.gemm_nt(a_, w, bias=bias, out=o_, split=SPLIT)
            e1.record(); torch.cuda.synchronize()
            t[v].append(e0.elapsed_time(e1) / REPS * 1e3)
    print(name + ": " + "  ".join(f"{v}: {min(x):6.1f} us ({2.0 * M * N * K / min(x) / 1e6:5.0f} TF/s)" for v, x in t.items()), flush=True)
os.environ.pop("SIMVG_GEMM_TALL", None)
