"""forward_test in a loop at batch size B (argv[1], default 1): for rocprofv3 --kernel-trace --stats"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from simvg_amd.models import build_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = build_model(bench.model_cfg()).to(dev).eval()
b = bench.synthetic_batch(B, 7, dev)
kw = dict(return_loss=False, with_bbox=True, with_mask=False, rescale=False)
with torch.no_grad():
    for _ in range(5):
        model(b["img"], b["ref_expr_inds"], b["img_metas"], text_attention_mask=b["text_attention_mask"], **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model(b["img"], b["ref_expr_inds"], b["img_metas"], text_attention_mask=b["text_attention_mask"], **kw)
        torch.cuda.synchronize()
    print(f"B={B}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per forward_test", flush=True)
