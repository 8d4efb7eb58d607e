"""GPU: BASELINE.json's FULL-size configurations -- config 2 (ViT-B, bs 64, one target) and configs 4/5 (ViT-L, bs 32,
GRefCOCO multi-target, num_queries 10) -- where the CPU oracle is too slow to be the checker (one ViT-L bs-32 training
step takes it minutes).  The oracle pins the kernels on the small fixtures (tests/test_model_gpu.py, batch 2-3); here the
same model is held to size-independent properties at the real batch, on the harsh (trained-scale) weights:

  P0  (round 4) the REFERENCE at full size: fixtures `base_nq1_full` / `large_nq10_grec_full` were recorded once from the
      executed reference (64 / 32 pairs, minutes of CPU and tens of GB of host memory in the dev container): the HIP path's
      boxes, the five losses and every parameter's gradient (norm + 16 entries) are compared with THOSE, in both precision
      modes; the exact-fp32 engine must sit on the reference (<= 5e-5 on boxes, 5e-5 on sampled gradient entries);
  P1  precision: the 16-bit engine's boxes against the reference's, L1 over normalised cxcywh, for ALL boxes of the batch
      (3 decoder layers x B x nq + B x nq token boxes).  Decoder branch (cross-attention averages the rounding noise of 400 image tokens): every box
      within the north_star bound of 1e-3 (measured max 6.1e-4).  Token branch (one object token's feature -> MLP -> box,
      no averaging): mean within 1e-3 (measured 4.7e-4 / 5.6e-4), maximum within 2e-3 (measured 1.25e-3: the maximum over
      hundreds of boxes of the deliberately harsh weights is a tail statistic -- the 2-3 pair fixtures measure <= 6.4e-4,
      the reference's own initialisation <= 7e-5).  `precision="fp32"` is the mode for a guaranteed 1e-3;
  P2  batch independence: a pair's boxes do not depend on what else is in the batch -- the first pairs of the full batch
      equal the same pairs run alone (different GEMM tile paths, same arithmetic to rounding);
  P3  padding: token ids stored at padded positions never reach the output (bit-exact);
  P4  training: one full step produces a finite loss, a finite gradient for every parameter except the unused
      `mask_token`, no fp16 saturation (no +-65504 in any 16-bit backward tensor that reaches a parameter gradient: the
      gradient norm of the 16-bit step agrees with the exact-fp32 step's), and gradients that point the same way as the
      exact-fp32 engine's (cosine over the whole encoder arena and over the head).
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
DEV = "cuda"

CASES = [("large", 32, 10, True), ("base", 64, 1, False)]
# recorded ONCE from the executed reference at these sizes (oracle/make_golden.py cases *_full: same weights seed 31, batch seed
# 77 as `_model` / `_batch` below): boxes, logits, the five losses, every parameter's gradient norm + 16 entries
FULL_FIXTURE = {("large", 32, 10): "large_nq10_grec_full", ("base", 64, 1): "base_nq1_full"}


def _model(vit, nq, seed=31):
    from oracle import ref_loader, simvg_cpu as O, weights as W
    from simvg_amd.models import build_model
    cfg = O.make_cfg(vit, nq, 640)
    mcfg = ref_loader.model_cfg(vit, nq, 640)
    model = build_model(mcfg)
    model.load_state_dict(W.golden_state_dict(cfg, seed), strict=True)
    return model.to(DEV), cfg


def _batch(cfg, B, grec, seed=77):
    from oracle import weights as W
    b = W.synthetic_batch(cfg, B, seed, grec)
    return dict(img=b["img"].to(DEV), ref_expr_inds=b["ref_expr_inds"].to(DEV), img_metas=b["img_metas"],
                text_attention_mask=b["text_attention_mask"].to(DEV), gt_bbox=[g.to(DEV) for g in b["gt_bbox"]])


def _boxes(model, b, sl=slice(None), ids=None):
    metas = b["img_metas"][sl]
    model(b["img"][sl], (b["ref_expr_inds"] if ids is None else ids)[sl], metas, return_loss=False,
          text_attention_mask=b["text_attention_mask"][sl], with_bbox=True, with_mask=False, rescale=False)
    out = model._last_output
    return {k: out[k].detach().float().clone() for k in ("outputs_coord_decoder_branch", "outputs_coord_token_branch")}


def _fp16():
    from simvg_amd import _lib
    return _lib.lowp_format() == "fp16"


def _tol():
    """Bounds of the 16-bit engine at full size.  fp16 build (shipped): the north_star's 1e-3 on every box of `forward_test`.
    bf16 build (BASELINE config 2's literal dtype; run by tests/test_bf16_build_gpu.py in a subprocess on libsimvg_hip_bf16.so):
    8 significand bits on trained-scale weights -- the bound test_model_gpu.py::_box_tol states for it (1.5e-2), losses 2e-2,
    gradient norms / entries as in test_model_gpu.py::_check_all_grads, direction cosine 0.97."""
    if _fp16():
        return dict(box=1e-3, train_tok=1e-3, alone=1e-3, loss=2e-3, gnorm=6e-2, gent=1.2e-1, cos=0.99, nratio=0.03)
    return dict(box=1.5e-2, train_tok=1.5e-2, alone=1.5e-2, loss=2e-2, gnorm=0.3, gent=0.6, cos=0.97, nratio=0.06)


def _l1(a, b):
    return float((a - b).abs().sum(-1).max())


def _l1_stats(a, b):
    d = (a - b).abs().sum(-1).reshape(-1).double()
    return float(d.max()), float(torch.quantile(d, 0.99)), float(d.mean()), d.numel()


@pytest.mark.parametrize("vit,B,nq,grec", CASES)
def test_full_size_inference_properties(golden, vit, B, nq, grec):
    fx = golden(FULL_FIXTURE[(vit, B, nq)])
    assert (fx["B"], fx["wseed"], fx["iseed"], fx["num_queries"]) == (B, 31, 77, nq)
    ref = {"outputs_coord_decoder_branch": fx["dec_boxes"].float(), "outputs_coord_token_branch": fx["tok_boxes"].float()}
    model, cfg = _model(vit, nq)
    model.eval()
    b = _batch(cfg, B, grec)
    T = _tol()
    with torch.no_grad():
        full = _boxes(model, b)
        # P3: garbage ids under the padding mask
        ids = b["ref_expr_inds"].clone()
        ids[b["text_attention_mask"] != 0] = 12345
        junk = _boxes(model, b, ids=ids)
        # P2: the first four pairs alone
        alone = _boxes(model, b, slice(0, 4))
        # P1: the exact-fp32 engine on the full batch
        model.vis_enc.set_precision("fp32")
        exact = _boxes(model, b)
    for k in full:
        assert torch.equal(full[k], junk[k]), ("padded ids reach the output", k)
        d2 = _l1(full[k][:, :4], alone[k])                                     # [layers, B, nq, 4]
        # the checker is the REFERENCE's own output at this size; the exact-fp32 engine is shown beside it (it must sit on the
        # reference: that is what makes it usable as the stand-in where no reference fixture exists)
        ex_mx = _l1_stats(exact[k].cpu(), ref[k])[0]
        assert ex_mx <= 5e-5, ("exact-fp32 engine vs the reference at full size", k, ex_mx)
        mx, p99, mean, n = _l1_stats(full[k].cpu(), ref[k])
        print(f"[full size {vit} B={B} nq={nq}] {k}: vs the REFERENCE over {n} boxes max {mx:.2e} p99 {p99:.2e} mean {mean:.2e} "
              f"(exact-fp32 engine vs the reference: max {ex_mx:.2e}); batch of {B} vs batch of 4: {d2:.2e}")
        # round 4: forward_test carries hi + lo 16-bit weights (`precise_inference`, simvg_gemm_nt_split); every box of both
        # branches of the full batch stays within the north_star's 1e-3 (round 3, single 16-bit weights: token max 1.11e-3 /
        # 1.17e-3 against the reference, mean 4.8e-4 / 5.2e-4).  bf16 build: `_tol`
        assert mx <= T["box"], (k, mx, p99, mean)
        assert d2 <= T["alone"], (k, d2)


def _grads(model):
    A = model.vis_enc._arena
    enc = A.flat_grad.detach().clone()
    head = torch.cat([p.grad.detach().float().reshape(-1) for n, p in model.named_parameters()
                      if not n.startswith("vis_enc.") and p.grad is not None])
    return enc, head


@pytest.mark.parametrize("vit,B,nq,grec", CASES)
def test_full_size_training_step_properties(golden, vit, B, nq, grec):
    fx = golden(FULL_FIXTURE[(vit, B, nq)])
    model, cfg = _model(vit, nq)
    model.eval()                  # dropout / DropPath off: the exact-fp32 engine has none, and the two must see one function
    b = _batch(cfg, B, grec)
    res = {}
    T = _tol()
    ref_boxes = {"outputs_coord_decoder_branch": fx["dec_boxes"].float(), "outputs_coord_token_branch": fx["tok_boxes"].float()}
    for prec in ("lowp", "fp32"):
        model.vis_enc.set_precision(prec)
        model.zero_grad(set_to_none=True)
        losses, _ = model(b["img"], b["ref_expr_inds"], b["img_metas"], return_loss=True,
                          text_attention_mask=b["text_attention_mask"], gt_bbox=b["gt_bbox"], rescale=False)
        losses["loss_total"].backward()
        torch.cuda.synchronize()
        if prec == "lowp":
            missing = [n for n, p in model.named_parameters() if p.grad is None]
            assert missing == ["vis_enc.beit3.vision_embed.mask_token"], missing
        res[prec] = (float(losses["loss_total"]),) + _grads(model)
        # against the reference's full-size step: the five losses, both whole-module gradient norms, every parameter's norm
        # and 16 sampled entries
        ltol, ntol, stol = (1e-4, 1e-3, 5e-5) if prec == "fp32" else (T["loss"], T["gnorm"], T["gent"])
        assert list(losses) == list(fx["losses"])
        # the boxes of the TRAINING forward at the full batch: every box of both branches within the north_star's 1e-3 (round 6:
        # `BEIT3.precise_training` -- hi + lo weights for the patch kernel and the qkv projection of the first third of the layers
        # (ViT-L: qkv of the first half + fc2 of the first quarter); measured max 8.25e-4 (ViT-B, 64 pairs) / 8.6e-4 (ViT-L, 32 x 10)
        # on the harsh weights; single 16-bit weights:
        # 1.11e-3 / 1.17e-3, the bound of round 5 was 1.25e-3).  These boxes feed the matcher and the losses (asserted right below)
        out = model._last_output
        for k, rb in ref_boxes.items():
            mx, p99, mean, n = _l1_stats(out[k].detach().float().cpu(), rb)
            bound = 5e-5 if prec == "fp32" else (T["box"] if "decoder" in k else T["train_tok"])
            print(f"[full size {vit} B={B} nq={nq}] {prec} TRAINING forward {k}: vs the REFERENCE over {n} boxes max {mx:.2e} p99 {p99:.2e} "
                  f"mean {mean:.2e} (bound {bound:g})")
            assert mx <= bound, (prec, k, mx, p99, mean)
        for k, v in fx["losses"].items():
            assert abs(float(losses[k]) - v) <= ltol * max(1.0, abs(v)), (prec, k, float(losses[k]), v)
        params = dict(model.named_parameters())
        from gradcheck import check_all_grads
        bad, worst_n, worst_s = check_all_grads(fx, params, ntol, stol, f"full size {vit} B={B} nq={nq} {prec}")
        print(f"[full size {vit} B={B} nq={nq}] {prec} step vs the REFERENCE's: losses within {ltol:g}; all "
              f"{len(fx['grads_all']['keys'])} parameter gradients: worst norm error {worst_n:.2e}, worst sampled entry "
              f"{worst_s:.2e} of the tensor's largest")
        assert not bad, (prec, bad[:12])
    (l16, e16, h16), (l32, e32, h32) = res["lowp"], res["fp32"]
    assert torch.isfinite(e16).all() and torch.isfinite(h16).all() and l16 == l16
    cos = lambda a, c: float(torch.dot(a.double(), c.double()) / (a.double().norm() * c.double().norm()))
    ce, ch = cos(e16, e32), cos(h16, h32)
    ne, nh = float(e16.norm() / e32.norm()), float(h16.norm() / h32.norm())
    print(f"[full size {vit} B={B} nq={nq}] loss 16-bit {l16:.5f} / fp32 {l32:.5f}; encoder gradient cosine {ce:.5f} "
          f"norm ratio {ne:.4f}; head gradient cosine {ch:.5f} norm ratio {nh:.4f}")
    assert abs(l16 - l32) <= T["loss"] * max(1.0, abs(l32))
    assert ce >= T["cos"] and ch >= T["cos"], (ce, ch)
    assert abs(ne - 1) <= T["nratio"] and abs(nh - 1) <= T["nratio"], (ne, nh)
