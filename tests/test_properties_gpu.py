"""Full-size (BASELINE.json configs[1]: ViT-B/32, 640x640, 20 tokens, B = 64) size-independent properties of the HIP
path -- the oracle finishes only small cases in seconds, so at this size the checks are structural:

  * batch independence: forward_test on 64 pairs == forward_test on the two halves (tile edges, row-group boundaries of
    the multiway GEMMs, attention workgroup indexing, head batching) -- bit-exact;
  * batch-permutation equivariance: permuting the pairs permutes the boxes -- bit-exact (every output element is the
    same dot product in the same k order whatever tile its row lands in);
  * determinism of the forward (no atomics in it): two runs are bit-identical;
  * linearity of the backward in the loss scale: grad(2 L) == 2 grad(L) up to the fp32 atomics' summation order;
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def full():
    import bench
    from simvg_amd.models import build_model
    torch.manual_seed(7)
    model = build_model(bench.model_cfg()).to("cuda")
    model.vis_enc._ensure_engine(torch.device("cuda"))
    batch = bench.synthetic_batch(64, 99, torch.device("cuda"))
    return model, batch


def _test_inputs(batch, idx=None):
    sel = (lambda t: t) if idx is None else (lambda t: t[idx])
    metas = batch["img_metas"] if idx is None else [batch["img_metas"][int(i)] for i in idx]
    return dict(img=sel(batch["img"]), ref_expr_inds=sel(batch["ref_expr_inds"]), img_metas=metas,
                text_attention_mask=sel(batch["text_attention_mask"]), return_loss=False, rescale=False)


def _boxes(model, **kw):
    with torch.no_grad():
        out = model(**kw)
    return out[0]["pred_bboxes"].clone(), out[1]["pred_bboxes"].clone()


def test_full_size_forward_is_deterministic_batch_independent_and_permutation_equivariant(full):
    model, batch = full
    model.eval()
    d0, t0 = _boxes(model, **_test_inputs(batch))
    d1, t1 = _boxes(model, **_test_inputs(batch))
    assert torch.equal(d0, d1) and torch.equal(t0, t1)                       # determinism
    assert torch.isfinite(d0).all() and (d0[:, 2] >= d0[:, 0]).all() and (d0[:, 3] >= d0[:, 1]).all()
    assert float(d0.min()) >= 0.0 and float(d0.max()) <= 640.0               # clipped to the image
    lo, hi = torch.arange(0, 32, device="cuda"), torch.arange(32, 64, device="cuda")
    dl, tl = _boxes(model, **_test_inputs(batch, lo))
    dh, th = _boxes(model, **_test_inputs(batch, hi))
    assert torch.equal(torch.cat([dl, dh]), d0) and torch.equal(torch.cat([tl, th]), t0)     # batch independence
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(3)).to("cuda")
    dp, tp = _boxes(model, **_test_inputs(batch, perm))
    assert torch.equal(dp, d0[perm]) and torch.equal(tp, t0[perm])            # permutation equivariance
    odd = torch.arange(0, 37, device="cuda")                                  # ragged batch: 37 pairs (partial tiles)
    do, to = _boxes(model, **_test_inputs(batch, odd))
    assert torch.equal(do, d0[:37]) and torch.equal(to, t0[:37])


def test_full_size_backward_is_linear_in_the_loss_scale(full):
    model, batch = full
    model.eval()                     # no dropout / DropPath: the two passes see the same function
    grads = []
    for scale in (1.0, 2.0):
        model.zero_grad(set_to_none=True)
        losses, _ = model(img=batch["img"], ref_expr_inds=batch["ref_expr_inds"], img_metas=batch["img_metas"],
                          text_attention_mask=batch["text_attention_mask"], gt_bbox=batch["gt_bbox"], rescale=False)
        (losses["loss_total"] * scale).backward()
        grads.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    g1, g2 = grads
    assert g1.keys() == g2.keys() and len(g1) > 300
    worst = 0.0
    for n in g1:
        ref = 2.0 * g1[n].float()
        den = float(ref.abs().max())
        if den == 0.0:
            assert float(g2[n].abs().max()) == 0.0, n
            continue
        worst = max(worst, float((g2[n].float() - ref).abs().max()) / den)
    # dgrad operands are rounded to bf16 AFTER the scale (bf16(2x) == 2 bf16(x) exactly), so only the summation order of
    # the fp32 atomics differs between the passes
    assert worst <= 2e-3, worst


def test_full_size_gradients_assign_mode_equals_accumulate_mode_and_accumulates(full, monkeypatch):
    """the first backward after zero_grad() WRITES the encoder Linears' weight gradients (second stage in assign mode, the arena
    skips their zero fill: simvg_amd/arena.py); (a) the same bits as with SIMVG_WGRAD_ASSIGN=0 (zero fill + accumulate) for the
    weights, the rest to the order of its atomics; (b) a second backward WITHOUT zero_grad() accumulates: gradients double."""
    model, batch = full
    model.eval()                     # no dropout / DropPath: every pass sees the same function

    def backward(zero=True):
        if zero:
            model.zero_grad(set_to_none=True)
        losses, _ = model(img=batch["img"], ref_expr_inds=batch["ref_expr_inds"], img_metas=batch["img_metas"],
                          text_attention_mask=batch["text_attention_mask"], gt_bbox=batch["gt_bbox"], rescale=False)
        losses["loss_total"].backward()
        return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    g_assign = backward()
    g_twice = backward(zero=False)
    monkeypatch.setenv("SIMVG_WGRAD_ASSIGN", "0")
    g_acc = backward()
    assert g_assign.keys() == g_acc.keys() == g_twice.keys() and len(g_assign) > 300
    lin = ("q_proj", "k_proj", "v_proj", "out_proj", "fc1", "fc2")
    n_lin = 0
    for n in g_assign:
        a, b, c = g_assign[n].float(), g_acc[n].float(), g_twice[n].float()
        den = float(b.abs().max())
        if den == 0.0:
            assert float(a.abs().max()) == 0.0 and float(c.abs().max()) == 0.0, n
            continue
        if "vis_enc" in n and n.endswith(".weight") and any(k in n for k in lin) and "encoder.layers" in n:
            assert torch.equal(a, b), n                  # slabs summed in a fixed order: bit-identical
            n_lin += 1
        else:
            assert float((a - b).abs().max()) / den <= 2e-3, n
        assert float((c - 2.0 * a).abs().max()) / den <= 4e-3, n
    assert n_lin >= 12 * 12


def test_other_geometry_512px_15_tokens_batch_independence_and_training_step():
    """the reference's 512x512 / max_token 15 dataset bases (configs/_base_/datasets/detection/*.py): 257 + 15 tokens"""
    import bench
    from simvg_amd.models import build_model
    torch.manual_seed(11)
    cfg = bench.model_cfg()
    cfg["vis_enc"]["img_size"] = 512
    cfg["head"]["text_max_token"] = 15
    model = build_model(cfg).to("cuda")
    B, S, T = 12, 512, 15
    g = torch.Generator().manual_seed(5)
    img = torch.randn(B, 3, S, S, generator=g).cuda()
    ids = torch.randint(4, 64010, (B, T), generator=g).cuda()
    ids[:, 0] = 0
    pad = torch.zeros(B, T, dtype=torch.int64).cuda()
    pad[:, 9:] = 1
    metas = [dict(img_shape=(S, S, 3), pad_shape=(S, S, 3), ori_shape=(S, S, 3), scale_factor=[1.0] * 4) for _ in range(B)]
    model.eval()
    kw = dict(return_loss=False, rescale=False)
    with torch.no_grad():
        full = model(img=img, ref_expr_inds=ids, img_metas=metas, text_attention_mask=pad, **kw)[0]["pred_bboxes"].clone()
        part = torch.cat([model(img=img[i:i + 5], ref_expr_inds=ids[i:i + 5], img_metas=metas[i:i + 5],
                                text_attention_mask=pad[i:i + 5], **kw)[0]["pred_bboxes"] for i in (0, 5, 10)])
    assert torch.equal(full, part) and torch.isfinite(full).all() and float(full.max()) <= 512.0
    model.train()
    gt = torch.tensor([[30.0, 40.0, 300.0, 350.0]]).repeat(B, 1).cuda()
    losses, _ = model(img=img, ref_expr_inds=ids, img_metas=metas, text_attention_mask=pad, gt_bbox=[b for b in gt], rescale=False)
    losses["loss_total"].backward()
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    assert torch.isfinite(losses["loss_total"]) and len(grads) > 300 and all(torch.isfinite(x).all() for x in grads)


@pytest.mark.parametrize("img_size,tokens", [(640, 1621), (480, 921)])
def test_other_geometry_patch16_against_the_exact_engine(img_size, tokens):
    """patch_size 16 (the BEiT-3 base checkpoint's native patch size): 1 + (S/16)^2 + 20 tokens per pair -- 1621 at 640x640 --
    beyond what one head's K / V can keep in LDS: the streamed-block attention kernels (online softmax over 256-key blocks).
    Checked against the exact-fp32 engine (its own fp32 attention kernels share no tiling with the 16-bit path): boxes within
    1e-3 L1, batch independence bit-exact, a finite training step; at 921 tokens (480x480 -- the exact backward keeps two
    [16, N] strips in LDS, N <= 1024) also the loss and the encoder gradient's direction and norm."""
    import bench
    from simvg_amd.models import build_model
    torch.manual_seed(13)
    cfg = bench.model_cfg()
    cfg["vis_enc"]["patch_size"] = 16
    cfg["vis_enc"]["img_size"] = img_size
    cfg["vis_enc"]["drop_path_rate"] = 0.0
    model = build_model(cfg).to("cuda")
    B, S = 3, img_size
    b = bench.synthetic_batch(B, 21, torch.device("cuda"))
    b["img"] = b["img"][:, :, :S, :S].contiguous()
    for m in b["img_metas"]:
        m.update(img_shape=(S, S, 3), pad_shape=(S, S, 3), ori_shape=(S, S, 3))
    b["gt_bbox"] = [g.clamp(max=float(S)) for g in b["gt_bbox"]]
    assert model.vis_enc.np + 1 + 20 == tokens
    model.eval()
    kw = dict(return_loss=False, rescale=False)

    def boxes(sl=slice(None)):
        with torch.no_grad():
            model(img=b["img"][sl], ref_expr_inds=b["ref_expr_inds"][sl], img_metas=b["img_metas"][sl],
                  text_attention_mask=b["text_attention_mask"][sl], **kw)
        out = model._last_output
        return torch.cat([out["outputs_coord_decoder_branch"][-1], out["outputs_coord_token_branch"][-1]], 1).float().clone()

    full = boxes()
    assert torch.equal(full[:2], boxes(slice(0, 2)))
    model.vis_enc.set_precision("fp32")
    exact = boxes()
    model.vis_enc.set_precision("lowp")
    l1 = float((full - exact).abs().sum(-1).max())
    res = {}
    for prec in (("lowp", "fp32") if tokens <= 1024 else ("lowp",)):
        model.vis_enc.set_precision(prec)
        model.zero_grad(set_to_none=True)
        losses, _ = model(img=b["img"], ref_expr_inds=b["ref_expr_inds"], img_metas=b["img_metas"],
                          text_attention_mask=b["text_attention_mask"], gt_bbox=b["gt_bbox"], rescale=False)
        losses["loss_total"].backward()
        res[prec] = (float(losses["loss_total"]), model.vis_enc._arena.flat_grad.detach().clone())
    l16, g16 = res["lowp"]
    print(f"[patch16, {tokens} tokens] boxes vs exact fp32 {l1:.2e}; loss {l16:.5f}")
    assert torch.isfinite(g16).all() and l16 == l16 and l1 <= 1e-3
    if "fp32" in res:
        l32, g32 = res["fp32"]
        cos = float(torch.dot(g16.double(), g32.double()) / (g16.double().norm() * g32.double().norm()))
        ratio = float(g16.norm() / g32.norm())
        print(f"[patch16, {tokens} tokens] loss fp32 {l32:.5f}; encoder gradient cosine {cos:.5f}, norm ratio {ratio:.4f}")
        assert abs(l16 - l32) <= 2e-3 * max(1.0, abs(l32)) and cos >= 0.99 and abs(ratio - 1) <= 0.03
