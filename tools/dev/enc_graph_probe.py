"""feasibility: the encoder's hand-sequenced forward / backward as hipGraphs -- host time of replay() while the GPU is busy"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from simvg_amd.models import build_model
from simvg_amd import hip_ops as ops

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = build_model(bench.model_cfg()).to(dev).train()
enc = model.vis_enc
enc.drop_path_probs = [0.0] * enc.L
enc._ensure_engine(dev)
ops.set_grad_scale(65536.0)
b = bench.synthetic_batch(64, 7, dev)
img, ids = b["img"], b["ref_expr_inds"]
pad = (b["text_attention_mask"] != 0).to(torch.uint8).contiguous()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        out, ws = enc._engine_forward(img, ids, pad, None, save=True)
        dout = torch.randn_like(out) * 1e-3
        enc._arena.begin_backward()
        enc._engine_backward(ws, dout, None)
    torch.cuda.synchronize()
    gf = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gf, stream=side):
        out, ws = enc._engine_forward(img, ids, pad, None, save=True)
    gb = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gb, stream=side):
        enc._engine_backward(ws, dout, None)
    torch.cuda.synchronize()
    for name, fn in (("eager", None), ("graph", (gf, gb))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        host_f = host_b = 0.0
        for _ in range(10):
            t = time.perf_counter()
            if fn is None:
                out, ws = enc._engine_forward(img, ids, pad, None, save=True)
            else:
                gf.replay()
            host_f += time.perf_counter() - t
            t = time.perf_counter()
            if fn is None:
                enc._engine_backward(ws, dout, None)
            else:
                gb.replay()
            host_b += time.perf_counter() - t
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        tw = time.perf_counter() - t0
        print(f"{name}: wall {tw / 10 * 1e3:.2f} ms per fwd+bwd; host returns after {th / 10 * 1e3:.2f} ms (fwd {host_f / 10 * 1e3:.2f}, bwd {host_b / 10 * 1e3:.2f})", flush=True)
