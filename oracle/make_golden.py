"""Generate tests/golden/*.pt from the REAL reference and pin the CPU restatement to it.

Run in the dev container only (needs /root/reference):
    python -m oracle.make_golden [--cases tiny_nq1,base_nq1,...]

For each case: build the reference MIXDETRMB (reference files executed verbatim through
oracle/ref_loader.py), load the seeded golden weights, run forward_test, forward_train
(eval-mode dropout / DropPath, SURVEY.md §7 hard part 5) and backward; run the restatement
oracle/simvg_cpu.py on the same state_dict and inputs and ASSERT agreement; then store the
reference's outputs as the fixture (inputs / weights are regenerated from the seeds).
"""
import argparse
import os
import sys
import time

import torch

from . import ref_loader, simvg_cpu as O, weights as W

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = {
    #  name            vit     nq  img  B  grec  wseed iseed
    "tiny_nq1":       ("tiny", 1, 128, 2, False, 11, 21),
    "tiny_nq10_grec": ("tiny", 10, 128, 3, True, 12, 22),
    "base_nq1":       ("base", 1, 640, 2, False, 13, 23),
    "base_nq10_grec": ("base", 10, 640, 3, True, 14, 24),
    "large_nq1":      ("large", 1, 640, 1, False, 15, 25),
    # BASELINE config 5: ViT-L + DWBD distillation, GRefCOCO multi-target (nq = 10, 2-target / no-target / 1-target samples)
    "large_nq10_grec":         ("large", 10, 640, 3, True, 18, 28),
    "large_nq10_grec_refinit": ("large", 10, 640, 3, True, 19, 29),
    # reference-style initialisation (what training from scratch / the benchmark uses)
    "base_nq1_refinit":       ("base", 1, 640, 2, False, 16, 26),
    "base_nq10_grec_refinit": ("base", 10, 640, 3, True, 17, 27),
    # branch_loss_weight = {"decoder": 1.0}: every *_twostage_1 config, pretrian-mixed / pretrain-cocoall (BASELINE config 4's
    # named workload) and the six finetune_* configs (e.g. configs/single/ViT-large/refcoco/refcoco_twostage_1.py:98,
    # configs/mix/ViT-base/pretrian-mixed.py:103); the token branch then runs forward only and gets no gradient
    "base_nq1_deconly":       ("base", 1, 640, 2, False, 31, 41, {"decoder": 1.0}),
    "large_nq1_deconly":      ("large", 1, 640, 2, False, 32, 42, {"decoder": 1.0}),
    "base_nq10_grec_deconly": ("base", 10, 640, 3, True, 33, 43, {"decoder": 1.0}),
    "tiny_nq1_deconly":       ("tiny", 1, 128, 2, False, 34, 44, {"decoder": 1.0}),
    # ViT-L's one-stage / two-stage-2 loss weights (configs/single/ViT-large/refcoco/refcoco_onestage.py:96)
    "large_nq1_w104":         ("large", 1, 640, 2, False, 35, 45, {"decoder": 1.0, "balanced_distill": {"token": 1.0, "distill": 0.4}}),
    "large_nq10_grec_w104_refinit": ("large", 10, 640, 3, True, 36, 46,
                                     {"decoder": 1.0, "balanced_distill": {"token": 1.0, "distill": 0.4}}),
    # BASELINE.json's FULL-size configurations, recorded from the executed reference once (minutes of CPU, tens of GB of host
    # memory): config 2 (ViT-B, 64 pairs) and configs 4 / 5 (ViT-L, 32 pairs, GRefCOCO multi-target, 10 queries), on the same
    # harsh weights (seed 31) and batch (seed 77) that tests/test_fullsize_gpu.py builds -- the HIP path is compared with THESE
    # boxes / logits / losses / per-parameter gradient norms, not with the repo's own exact-fp32 engine
    "base_nq1_full":          ("base", 1, 640, 64, False, 31, 77),
    "large_nq10_grec_full":   ("large", 10, 640, 32, True, 31, 77),
}

GRAD_KEYS = [  # sampled gradient probes (first 16 elements + norm)
    "vis_enc.beit3.vision_embed.proj.weight", "vis_enc.beit3.vision_embed.cls_token",
    "vis_enc.beit3.encoder.embed_positions.A.weight", "vis_enc.beit3.encoder.embed_positions.B.weight",
    "vis_enc.beit3.encoder.layers.0.self_attn.q_proj.A.weight", "vis_enc.beit3.encoder.layers.0.self_attn.k_proj.B.weight",
    "vis_enc.beit3.encoder.layers.0.self_attn.v_proj.A.bias", "vis_enc.beit3.encoder.layers.0.self_attn.out_proj.B.weight",
    "vis_enc.beit3.encoder.layers.0.self_attn.inner_attn_ln.A.weight", "vis_enc.beit3.encoder.layers.0.self_attn_layer_norm.B.bias",
    "vis_enc.beit3.encoder.layers.1.ffn.A.fc1.weight", "vis_enc.beit3.encoder.layers.1.ffn.B.fc2.weight",
    "vis_enc.beit3.encoder.layers.1.ffn.A.ffn_layernorm.weight", "vis_enc.beit3.encoder.layers.1.final_layer_norm.A.weight",
    "vis_enc.beit3.encoder.layer_norm.A.weight", "vis_enc.beit3.encoder.layer_norm.B.bias",
    "head.input_proj.weight", "head.input_text_proj.weight", "head.input_cls_proj.bias", "head.query_embed.weight",
    "head.mlp.layers.0.weight", "head.class_embed_token.weight", "head.bbox_embed_token.layers.2.weight",
    "head.class_embed_decoder.bias", "head.bbox_embed_decoder.layers.0.weight",
    "head.transformer.decoder.layers.0.attentions.1.attn.in_proj_weight",
    "head.transformer.decoder.layers.2.attentions.0.attn.out_proj.weight",
    "head.transformer.decoder.layers.1.ffns.0.layers.0.0.weight", "head.transformer.decoder.layers.2.norms.1.weight",
    "head.transformer.decoder.post_norm_layer.weight",
    "head.text_guided_query_generation_transformer.layers.1.attentions.1.attn.in_proj_weight",
    "head.text_guided_query_generation_transformer.layers.0.ffns.0.layers.1.weight",
]


def _even_idx(numel, n):
    """n evenly spaced flat indices (float64: float32's 24 bits run out at the 49 M-entry text table)"""
    return torch.linspace(0, numel - 1, n, dtype=torch.float64).floor().long().clamp_(0, numel - 1)


def _all_grads(ref_grads, n=16):
    """EVERY parameter's gradient: norm + n evenly spaced entries, packed into four tensors (a few tens of kB per fixture)"""
    keys = [k for k, g in ref_grads.items() if g is not None]
    idx = torch.zeros(len(keys), n, dtype=torch.int32)
    vals = torch.zeros(len(keys), n)
    norm = torch.zeros(len(keys), dtype=torch.float64)
    amax = torch.zeros(len(keys))
    for i, k in enumerate(keys):
        t = ref_grads[k].detach().float().reshape(-1)
        ix = _even_idx(t.numel(), n) if t.numel() >= n else torch.arange(n) % t.numel()
        idx[i], vals[i] = ix.int(), t[ix]
        norm[i] = float(t.double().norm())
        amax[i] = float(t.abs().max())
    return dict(keys=keys, idx=idx, vals=vals, norm=norm, amax=amax)


def _summ(t, n=64):
    t = t.detach().float().reshape(-1)
    idx = _even_idx(t.numel(), min(n, t.numel()))
    return dict(sum=float(t.double().sum()), abssum=float(t.double().abs().sum()), max=float(t.abs().max()),
                idx=idx, vals=t[idx].clone())


def build_reference(vit, nq, img_size, cfg, blw=None):
    M = ref_loader.load()
    mcfg = ref_loader.model_cfg("base" if vit == "tiny" else vit, nq, img_size, cfg.patch_size, blw)
    if vit == "tiny":
        beit3_mod = sys.modules["simvg.models.vis_encs.beit.beit3"]
        from .leaf import EncoderConfig
        orig = beit3_mod._get_base_config
        beit3_mod._get_base_config = lambda img_size, patch_size, drop_path_rate, vocab_size: EncoderConfig(
            img_size=img_size, patch_size=patch_size, vocab_size=vocab_size, multiway=True,
            layernorm_embedding=False, normalize_output=True, no_output_layer=True, drop_path_rate=0.0,
            encoder_embed_dim=cfg.embed_dim, encoder_attention_heads=cfg.heads,
            encoder_ffn_embed_dim=cfg.ffn_dim, encoder_layers=cfg.layers)
        mcfg["head"]["in_channels"] = cfg.embed_dim
        try:
            return M.build_model(mcfg)
        finally:
            beit3_mod._get_base_config = orig
    return M.build_model(mcfg)


def run_case(name, check_only=False):
    vit, nq, img_size, B, grec, wseed, iseed = CASES[name][:7]
    blw = CASES[name][7] if len(CASES[name]) > 7 else None
    cfg = O.cfg_from_branch_loss_weight(O.make_cfg(vit, nq, img_size), blw)
    t0 = time.time()
    model = build_reference(vit, nq, img_size, cfg, blw)
    refinit = name.endswith("_refinit")
    sd = W.reference_init_state_dict(cfg, wseed) if refinit else W.golden_state_dict(cfg, wseed)
    missing = model.load_state_dict(sd, strict=True)
    batch = W.synthetic_batch(cfg, B, iseed, grec)
    model.eval()   # dropout / DropPath identity; losses are still computed by forward_train
    kw = dict(text_attention_mask=batch["text_attention_mask"])
    # ---------------- reference forward_test ----------------
    metas = [dict(m) for m in batch["img_metas"]]
    pred = model(batch["img"], batch["ref_expr_inds"], metas, return_loss=False, with_bbox=True,
                 with_mask=False, rescale=False, **kw)
    # ---------------- reference forward_train + backward ----------------
    model.zero_grad()
    metas = [dict(m) for m in batch["img_metas"]]
    B_, _, H_, W_ = batch["img"].shape
    model.add_batch_input_shape(batch["img"], metas)
    img_feat, text_feat, cls_feat = model.extract_visual_language(batch["img"], batch["ref_expr_inds"],
                                                                  batch["text_attention_mask"])
    x_mm = img_feat.transpose(-1, -2).reshape(B_, -1, H_ // cfg.patch_size, W_ // cfg.patch_size)
    losses, hout = model.head.forward_train(x_mm, metas, cls_feat=cls_feat, gt_bbox=batch["gt_bbox"],
                                            text_feat=text_feat, text_mask=batch["text_attention_mask"])
    losses["loss_total"].backward()
    ref_grads = {k: p.grad for k, p in model.named_parameters()}
    print(f"[{name}] reference done in {time.time() - t0:.1f}s; losses",
          {k: round(float(v), 6) for k, v in losses.items()})
    # ---------------- restatement on the same state_dict ----------------
    sdg = {k: v.clone().requires_grad_(v.is_floating_point() and "empty_weight" not in k) for k, v in sd.items()}
    metas2 = [dict(m) for m in batch["img_metas"]]
    pred2, _ = O.forward_test(sd, cfg, batch["img"], batch["ref_expr_inds"], metas2, batch["text_attention_mask"])
    metas2 = [dict(m) for m in batch["img_metas"]]
    losses2, out2, detail2 = O.forward_train(sdg, cfg, batch["img"], batch["ref_expr_inds"], metas2,
                                             batch["text_attention_mask"], batch["gt_bbox"])
    losses2["loss_total"].backward()

    def chk(a, b, what, tol=2e-5):
        if a is None or b is None:       # the decoder-only head has no token branch: both sides must say so
            assert a is None and b is None, f"[{name}] {what}: one side is None"
            return 0.0
        err = float((a.detach() - b.detach()).abs().max()) if a.numel() else 0.0
        scale = max(1.0, float(b.detach().abs().max())) if b.numel() else 1.0
        assert err <= tol * scale, f"[{name}] restatement mismatch on {what}: {err}"
        return err

    errs = {}
    errs["img_feat"] = chk(O.beit3_forward(sd, cfg, batch["img"], batch["ref_expr_inds"], batch["text_attention_mask"])[0], img_feat, "img_feat")
    errs["tok_logits"] = chk(out2["tok_logits"], hout["outputs_class_token_branch"], "tok_logits")
    errs["tok_boxes"] = chk(out2["tok_boxes"], hout["outputs_coord_token_branch"], "tok_boxes")
    errs["dec_logits"] = chk(out2["dec_logits"], hout["outputs_class_decoder_branch"], "dec_logits")
    errs["dec_boxes"] = chk(out2["dec_boxes"], hout["outputs_coord_decoder_branch"], "dec_boxes")
    assert list(losses2) == list(losses), (list(losses2), list(losses))      # same keys, same (insertion) order
    for k in losses:
        errs[k] = chk(losses2[k], losses[k], k)
    if not grec:
        for i in (0, 1):
            errs[f"pred{i}"] = chk(pred2[i]["pred_bboxes"], pred[i]["pred_bboxes"], f"pred_bboxes[{i}]", 1e-4)
    else:
        for i in (0, 1):
            if pred[i]["pred_bboxes"] is None or pred2[i]["pred_bboxes"] is None:
                assert pred[i]["pred_bboxes"] is None and pred2[i]["pred_bboxes"] is None, i
                continue
            for a, b in zip(pred2[i]["pred_bboxes"], pred[i]["pred_bboxes"]):
                chk(a["boxes"], b["boxes"], "grec boxes", 1e-4)
                chk(a["scores"], b["scores"], "grec scores")
    gerr = 0.0
    for k, g in ref_grads.items():
        g2 = sdg[k].grad
        if g is None:
            assert g2 is None or float(g2.abs().max()) == 0.0, k
            continue
        assert g2 is not None, f"[{name}] the reference has a gradient for {k}, the restatement has none"
        gerr = max(gerr, chk(g2, g, "grad " + k, 5e-5))
    errs["grad_max"] = gerr
    print(f"[{name}] restatement == reference; max errs:", {k: f"{v:.2e}" for k, v in errs.items()})
    if check_only:
        return
    # ---------------- fixture ----------------
    tiny = vit == "tiny"
    fx = dict(
        name=name, refinit=refinit, vit=vit, num_queries=nq, img_size=img_size, B=B, grec=grec, wseed=wseed, iseed=iseed,
        torch_version=torch.__version__, branch_loss_weight=blw,
        no_grad_params=sorted(k for k, g in ref_grads.items() if g is None),
        losses={k: float(v) for k, v in losses.items()},
        tok_logits=None if hout["outputs_class_token_branch"] is None else hout["outputs_class_token_branch"].detach().clone(),
        tok_boxes=None if hout["outputs_coord_token_branch"] is None else hout["outputs_coord_token_branch"].detach().clone(),
        token_features=_summ(hout["token_features"], 512),
        dec_logits=hout["outputs_class_decoder_branch"].detach().clone(),
        dec_boxes=hout["outputs_coord_decoder_branch"].detach().clone(),
        img_feat=img_feat.detach().clone() if tiny else _summ(img_feat, 4096),
        text_feat=text_feat.detach().clone() if tiny else _summ(text_feat, 2048),
        cls_feat=cls_feat.detach().clone(),
        matcher_gt=[(a.clone(), b.clone()) for a, b in
                    model.head.matcher(hout["decoder_branch_output"], detail2["targets_gt"])],
        grad_norm_vis_enc=float(torch.sqrt(sum((g.double() ** 2).sum() for k, g in ref_grads.items()
                                               if g is not None and k.startswith("vis_enc")))),
        grad_norm_head=float(torch.sqrt(sum((g.double() ** 2).sum() for k, g in ref_grads.items()
                                            if g is not None and k.startswith("head")))),
        grads={k: dict(norm=float(ref_grads[k].norm()), head=ref_grads[k].reshape(-1)[:16].clone(),
                       summ=_summ(ref_grads[k], 64)) for k in GRAD_KEYS if k in ref_grads and ref_grads[k] is not None},
        grads_all=_all_grads(ref_grads),
    )
    for i, key in enumerate(["pred_decoder", "pred_token"]):
        pb = pred[i]["pred_bboxes"]
        if pb is None:
            fx[key] = None
        elif not grec:
            fx[key] = pb.clone()
        else:
            fx[key] = [{k: v.clone() for k, v in d.items()} for d in pb]
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, name + ".pt")
    torch.save(fx, path)
    print(f"[{name}] wrote {path} ({os.path.getsize(path) / 1024:.1f} kB)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="tiny_nq1,tiny_nq10_grec,base_nq1,base_nq10_grec")
    ap.add_argument("--check-only", action="store_true")
    a = ap.parse_args()
    torch.manual_seed(0)
    for c in a.cases.split(","):
        run_case(c, a.check_only)


if __name__ == "__main__":
    main()
