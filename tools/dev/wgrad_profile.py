"""Role timeline of the XCD-partitioned wgrad (library built with -DSIMVG_WG_PROFILE): s_memtime (shader cycles) of the first wave of each
role group in workgroup 8, stages 8..23.  Samples: 0 top of LOADa, 1 reads issued, 2 after barrier A, 3 DMA issued, 4 LOADb done,
5 after barrier B, 6 MFMAs issued, 7 after barrier C."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
prof = torch.zeros(3 * 16 * 8, dtype=torch.int64, device="cuda")
os.environ["SIMVG_WG_PROF_PTR"] = str(prof.data_ptr())
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
names = ["reads issue", "wait+barA", "DMA issue", "LOADb rest", "wait+barB", "MFMA issue", "wait+barC"]
for g in range(3):
    d = p[g, :, 1:] - p[g, :, :-1]
    per_stage = p[g, 1:, 0] - p[g, :-1, 0]
    print(f"group {g}: stage period mean {per_stage.mean():.0f} cycles; " + "  ".join(f"{n} {d[:, i].mean():.0f}" for i, n in enumerate(names)))
t0 = p[:, 0, 0].min()
for t in range(3):
    print("stage", 8 + t, " ".join(f"g{g}:" + ",".join(f"{p[g, t, k] - t0:.0f}" for k in range(8)) for g in range(3)))
