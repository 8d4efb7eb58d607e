"""Print the parity error breakdown of the HIP model vs the reference fixtures (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_model_gpu import _build, _dev_batch

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
for name in sys.argv[1:] or ["tiny_nq1", "tiny_nq10_grec", "base_nq1", "base_nq10_grec", "large_nq1"]:
    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    model, batch, cfg = _build(fx)
    model.eval()
    db = _dev_batch(batch)
    losses, preds = model(db["img"], db["ref_expr_inds"], db["img_metas"], return_loss=True,
                          text_attention_mask=db["text_attention_mask"], gt_bbox=batch["gt_bbox"])
    out = model._last_output
    r = {}
    for key, fkey in [("outputs_coord_decoder_branch", "dec_boxes"), ("outputs_coord_token_branch", "tok_boxes")]:
        d = (out[key].detach().float().cpu() - fx[fkey]).abs().sum(-1)
        r[fkey + "_L1max"] = float(d.max()); r[fkey + "_L1mean"] = float(d.mean())
    for key, fkey in [("outputs_class_decoder_branch", "dec_logits"), ("outputs_class_token_branch", "tok_logits")]:
        r[fkey + "_abs"] = float((out[key].detach().float().cpu() - fx[fkey]).abs().max())
        r[fkey + "_scale"] = float(fx[fkey].abs().max())
    enc = model.vis_enc
    ws = list(enc._ws.values())[-1]
    B, T = db["ref_expr_inds"].shape
    img_feat, text_feat, cls_feat = enc.split_output(ws["out"], B, T)
    r["cls_feat_abs"] = float((cls_feat.float().cpu() - fx["cls_feat"]).abs().max()); r["cls_scale"] = float(fx["cls_feat"].abs().max())
    r["losses"] = {k: (round(float(losses[k]), 5), round(v, 5)) for k, v in fx["losses"].items()}
    print(name, {k: (round(v, 6) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
    if os.environ.get("DIAG_GRADS"):
        model.zero_grad(set_to_none=True)
        losses["loss_total"].backward()
        params = dict(model.named_parameters())
        for k, gp in fx["grads"].items():
            g = params[k].grad
            ref = gp["summ"]
            got = g.detach().float().cpu().reshape(-1)[ref["idx"]]
            e = float((got - ref["vals"]).norm()) / max(float(ref["vals"].norm()), 1e-12)
            cos = float((got * ref["vals"]).sum() / (got.norm() * ref["vals"].norm() + 1e-20))
            print(f"   {k:90s} relL2 {e:.4f} cos {cos:.4f} norm {float(g.norm()):.4e} ref {gp['norm']:.4e} samp_norm/ref_norm {float(ref['vals'].norm())/gp['norm']:.3f}")
