"""Phase timeline of the one-barrier wgrad variant (SIMVG_WGRAD=x1, library built with -DSIMVG_WG_PROFILE): s_memtime of the first wave
of each group in workgroup 8, stages 8..23.  Stamps: 0 top, 1 reads issued, 2 MFMAs issued, 3 DMA issued, 4 vmcnt wait done, 5 barrier passed.
Group orders: g0 R M D, g1 D R M, g2 R D M."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
prof = torch.zeros(3 * 16 * 8, dtype=torch.int64, device="cuda")
os.environ["SIMVG_WG_PROF_PTR"] = str(prof.data_ptr())
os.environ["SIMVG_WGRAD"] = "x1"
from simvg_amd import hip_ops as ops
M, SPLIT = 26944, 25664
N, K = {"fc1": (3072, 768), "qkv": (2304, 768), "fc2": (768, 3072)}[sys.argv[1] if len(sys.argv) > 1 else "fc1"]
a = torch.randn(M, K, device="cuda").to(ops.LP())
dy = torch.randn(M, N, device="cuda").to(ops.LP())
dw = torch.zeros(2, N, K, device="cuda")
db = torch.zeros(2, N, device="cuda")
for _ in range(20):
    ops.gemm_tn(dy, a, dw, split=SPLIT, db=db)
torch.cuda.synchronize()
p = prof.cpu().view(3, 16, 8).double()
order = {0: [0, 1, 2, 3, 4, 5], 1: [0, 3, 1, 2, 4, 5], 2: [0, 1, 3, 2, 4, 5]}
names = {1: "reads", 2: "mfma", 3: "dma", 4: "vmwait", 5: "barrier"}
t0 = p[:, 0, 0].min()
for g in range(3):
    o = order[g]
    per_stage = (p[g, 1:, 0] - p[g, :-1, 0]).mean()
    segs = []
    for a_, b_ in zip(o[:-1], o[1:]):
        segs.append(f"{names[b_]} {(p[g, :, b_] - p[g, :, a_]).mean():.0f}")
    gap = (p[g, 1:, 0] - p[g, :-1, 5]).mean()
    print(f"group {g}: stage period {per_stage:.0f} cycles; " + "  ".join(segs) + f"  loop-back {gap:.0f}")
for t in range(2):
    print("stage", 8 + t, " | ".join(f"g{g}: " + ",".join(f"{names.get(k, 'top')}@{p[g, t, k] - t0:.0f}" for k in order[g]) for g in range(3)))
