"""Every exact-fp32 GEMM launch (gemm_f32 / gemm_f32_group) of ONE training step with its problems' shapes, operand orientation and
event time: python tools/dev/head_gemm_census.py [num_queries=10] [batch=64]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from simvg_amd import hip_ops as ops
from simvg_amd.models import build_model

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = build_model(bench.model_cfg(nq, "base")).to(dev).train()
batch = bench.synthetic_batch(B, 1000, dev)
rec, on = [], [False]
orig_group, orig_one = ops.gemm_f32_group, ops.gemm_f32


def ori(sk, sother):
    return "K" if sk == 1 else "N"


def group(problems):
    if not on[0]:
        return orig_group(problems)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig_group(problems)
    e1.record()
    rec.append((e0, e1, tuple((q["M"], q["N"], q["K"], ori(q["sak"], q["sam"]) + ori(q["sbk"], q["sbn"]),
                                "+".join(k for k in ("bias", "addend", "A2", "B2", "mult", "gate") if q.get(k) is not None) + (",acc" if q.get("accumulate") else "") + f",act{q.get('act', 0)}") for q in problems)))
    return r


def one(A, sam, sak, Bm, sbk, sbn, C, M, N, K, **kw):
    if not on[0]:
        return orig_one(A, sam, sak, Bm, sbk, sbn, C, M, N, K, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig_one(A, sam, sak, Bm, sbk, sbn, C, M, N, K, **kw)
    e1.record()
    rec.append((e0, e1, ((M, N, K, ori(sak, sam) + ori(sbk, sbn), "single"),)))
    return r


ops.gemm_f32_group, ops.gemm_f32 = group, one
import simvg_amd.models.heads.functions as F
for mod in (F,):
    if hasattr(mod, "ops"):
        mod.ops.gemm_f32_group, mod.ops.gemm_f32 = group, one


def step():
    losses, _ = model(batch["img"], batch["ref_expr_inds"], batch["img_metas"], return_loss=True,
                      text_attention_mask=batch["text_attention_mask"], gt_bbox=batch["gt_bbox"], rescale=False)
    model.zero_grad(set_to_none=True)
    losses["loss_total"].backward()


for _ in range(3):
    step()
on[0] = True
step()
torch.cuda.synchronize()
tot = 0.0
agg = collections.OrderedDict()
for e0, e1, probs in rec:
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    d = agg.setdefault(probs, [0, 0.0])
    d[0] += 1
    d[1] += us
print(f"{len(rec)} launches, {tot:.0f} us (event-bracketed: each includes ~2 us of bracket)")
for probs, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:3d} x {us / n:7.1f} us = {us:7.0f}   " + " | ".join(f"{m}x{nn}x{k} {o} {x}" for m, nn, k, o, x in probs))
