"""CPU: the oracle restatement (oracle/simvg_cpu.py) reproduces the fixtures that
oracle/make_golden.py recorded from the REAL reference files (tests/golden/*.pt).
Tolerance: fp32 vs fp32 on the same torch build -> 5e-5 relative to max|ref|."""
import pytest
import torch

from oracle import simvg_cpu as O, weights as W


def _close(a, b, tol=5e-5):
    scale = max(1.0, float(b.abs().max())) if b.numel() else 1.0
    err = float((a - b).abs().max()) if b.numel() else 0.0
    assert err <= tol * scale, f"err {err} scale {scale}"


def _check_summ(t, s, tol=5e-5):
    t = t.detach().float().reshape(-1)
    _close(t[s["idx"]], s["vals"], tol)
    assert abs(float(t.double().abs().sum()) - s["abssum"]) <= 1e-4 * max(1.0, s["abssum"])


def _run(fx, backward):
    cfg = O.cfg_from_branch_loss_weight(O.make_cfg(fx["vit"], fx["num_queries"], fx["img_size"]), fx.get("branch_loss_weight"))
    sd = W.golden_state_dict(cfg, fx["wseed"])
    batch = W.synthetic_batch(cfg, fx["B"], fx["iseed"], fx["grec"])
    if backward:
        sd = {k: v.clone().requires_grad_(v.is_floating_point() and "empty_weight" not in k) for k, v in sd.items()}
    losses, out, detail = O.forward_train(sd, cfg, batch["img"], batch["ref_expr_inds"], batch["img_metas"],
                                          batch["text_attention_mask"], batch["gt_bbox"])
    return cfg, sd, batch, losses, out, detail


@pytest.mark.parametrize("name", ["tiny_nq1", "tiny_nq10_grec", "tiny_nq1_deconly"])
def test_tiny_full(golden, name):
    fx = golden(name)
    cfg, sd, batch, losses, out, detail = _run(fx, backward=True)
    for k in ["tok_logits", "tok_boxes", "dec_logits", "dec_boxes"]:
        if fx[k] is None:          # branch_loss_weight={"decoder": 1.0}: the reference's head has no token branch
            assert out[k] is None, k
            continue
        _close(out[k].detach(), fx[k])
    assert list(losses) == list(fx["losses"])
    for k, v in fx["losses"].items():
        assert abs(float(losses[k]) - v) <= 5e-5 * max(1.0, abs(v)), k
    img_feat, text_feat, cls_feat = O.beit3_forward({k: v.detach() for k, v in sd.items()}, cfg, batch["img"],
                                                    batch["ref_expr_inds"], batch["text_attention_mask"])
    _close(img_feat, fx["img_feat"])
    _close(text_feat, fx["text_feat"])
    _close(cls_feat, fx["cls_feat"])
    losses["loss_total"].backward()
    for k, g in fx["grads"].items():
        _close(sd[k].grad.reshape(-1)[:16], g["head"], 1e-4)
        assert abs(float(sd[k].grad.norm()) - g["norm"]) <= 1e-4 * max(1.0, g["norm"]), k
    for k in fx.get("no_grad_params", []):     # parameters the reference's backward never reaches
        assert sd[k].grad is None or float(sd[k].grad.abs().max()) == 0.0, k
    # matcher indices recorded from the reference's own HungarianMatcher call
    idx = O.hungarian(out["dec_logits"][-1].detach(), out["dec_boxes"][-1].detach(), detail["targets_gt"], cfg)
    for (a, b), (ra, rb) in zip(idx, fx["matcher_gt"]):
        assert torch.equal(a, ra) and torch.equal(b, rb)


@pytest.mark.parametrize("name", ["tiny_nq1", "tiny_nq10_grec", "tiny_nq1_deconly"])
def test_tiny_forward_test(golden, name):
    fx = golden(name)
    cfg = O.cfg_from_branch_loss_weight(O.make_cfg(fx["vit"], fx["num_queries"], fx["img_size"]), fx.get("branch_loss_weight"))
    sd = W.golden_state_dict(cfg, fx["wseed"])
    batch = W.synthetic_batch(cfg, fx["B"], fx["iseed"], fx["grec"])
    pred, _ = O.forward_test(sd, cfg, batch["img"], batch["ref_expr_inds"], batch["img_metas"], batch["text_attention_mask"])
    if not fx["grec"]:
        _close(pred[0]["pred_bboxes"], fx["pred_decoder"], 1e-4)
        if fx["pred_token"] is None:
            assert pred[1] == dict(pred_bboxes=None, pred_masks=None, predict_classes=None)
        else:
            _close(pred[1]["pred_bboxes"], fx["pred_token"], 1e-4)
    else:
        for i, key in enumerate(["pred_decoder", "pred_token"]):
            for a, b in zip(pred[i]["pred_bboxes"], fx[key]):
                _close(a["boxes"], b["boxes"], 1e-4)
                _close(a["scores"], b["scores"])
                assert torch.equal(a["labels"], b["labels"])


@pytest.mark.slow
def test_base_nq1_forward(golden):
    """Full-size ViT-B/32 @640 (N=421) forward + losses against the reference fixture."""
    fx = golden("base_nq1")
    with torch.no_grad():
        cfg, sd, batch, losses, out, _ = _run(fx, backward=False)
    for k in ["tok_logits", "tok_boxes", "dec_logits", "dec_boxes"]:
        _close(out[k], fx[k])
    for k, v in fx["losses"].items():
        assert abs(float(losses[k]) - v) <= 5e-5 * max(1.0, abs(v)), k
    img_feat, text_feat, cls_feat = O.beit3_forward(sd, cfg, batch["img"], batch["ref_expr_inds"], batch["text_attention_mask"])
    _check_summ(img_feat, fx["img_feat"])
    _check_summ(text_feat, fx["text_feat"])
    _close(cls_feat, fx["cls_feat"])


def test_quirk_q1_int64_mask_vs_bool():
    """Q1: `~text_mask` on the int64 mask is a bitwise NOT -> integer indexing (rows -1/-2)."""
    cfg = O.make_cfg("tiny", 1, 128)
    torch.manual_seed(0)
    feat = torch.randn(20, 8)
    m_int = torch.tensor([0] * 5 + [1] * 15)
    got = torch.max(feat[~m_int, :], dim=0)[0]
    assert torch.equal(got, torch.maximum(feat[19], feat[18]))
    got_b = torch.max(feat[~m_int.bool(), :], dim=0)[0]
    assert torch.equal(got_b, feat[:5].max(0)[0])


def test_quirk_q2_sine1d():
    e = O.sine_pos_1d(20, 256)
    assert e.shape == (20, 256)
    pos = torch.arange(20, dtype=torch.float)
    assert torch.allclose(e[:, 0], torch.sin(pos)) and torch.allclose(e[:, 1], torch.cos(pos))
    assert float(e[:, 2::2].abs().max()) == 0.0 and float((e[:, 3::2] - 1).abs().max()) == 0.0
