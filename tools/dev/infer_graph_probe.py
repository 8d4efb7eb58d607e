"""feasibility: forward_test of the whole model as ONE hipGraph (static inputs), latency vs eager"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from simvg_amd.models import build_model

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = build_model(bench.model_cfg()).to(dev).eval()
for B in (1, 8):
    b = bench.synthetic_batch(B, 7, dev)
    kw = dict(return_loss=False, with_bbox=True, with_mask=False, rescale=False)
    static = dict(img=b["img"].clone(), ids=b["ref_expr_inds"].clone(), mask=b["text_attention_mask"].clone())

    def run():
        with torch.no_grad():
            return model(static["img"], static["ids"], b["img_metas"], text_attention_mask=static["mask"], **kw)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            ref = run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = run()
    torch.cuda.synchronize()
    # new input
    b2 = bench.synthetic_batch(B, 8, dev)
    static["img"].copy_(b2["img"]); static["ids"].copy_(b2["ref_expr_inds"]); static["mask"].copy_(b2["text_attention_mask"])
    g.replay()
    torch.cuda.synchronize()
    with torch.no_grad():
        eager = model(b2["img"], b2["ref_expr_inds"], b2["img_metas"], text_attention_mask=b2["text_attention_mask"], **kw)
    same = all(torch.equal(o["pred_bboxes"], e["pred_bboxes"]) for o, e in zip(out, eager))
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
        torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / 50 * 1e3
    t0 = time.perf_counter()
    for _ in range(50):
        with torch.no_grad():
            model(b2["img"], b2["ref_expr_inds"], b2["img_metas"], text_attention_mask=b2["text_attention_mask"], **kw)
        torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / 50 * 1e3
    print(f"B={B}: graph {tg:.3f} ms, eager {te:.3f} ms, identical boxes: {same}", flush=True)
