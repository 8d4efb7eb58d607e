"""CPU restatement of the SimVG hot path (TEST INFRASTRUCTURE -- the parity oracle).

Pure-PyTorch fp32, functional over a ``state_dict`` that uses the reference's own key
schema, so the same dict drives the reference, this oracle and the
HIP product.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; ``simvg_amd`` never does.

Pinning: every function here is checked against the REAL reference files executed from
/root/reference (``oracle/ref_loader.py`` + ``oracle/make_golden.py``; fixtures under
``tests/golden/``).  The third-party leaf arithmetic (torchscale / detrex / detectron2,
not vendored in the reference and absent from this image) is restated in ``oracle/leaf.py``
and is PARITY UNPINNED by the reference itself (it has no tests) -- see that file's header.

Reference lines followed by each function are cited inline (paths relative to
/root/reference).
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from .leaf import (box_cxcywh_to_xyxy, box_iou, box_xyxy_to_cxcywh, generalized_box_iou)


# ----------------------------------------------------------------------------------------
# configuration (configs/single/ViT-base/refcoco/refcoco_onestage.py:68-105 +
# simvg/models/vis_encs/beit/modeling_utils.py:21-70)
# ----------------------------------------------------------------------------------------
def make_cfg(vit_type="base", num_queries=1, img_size=640, patch_size=32, max_token=20, **over):
    if vit_type == "base":
        enc = dict(embed_dim=768, heads=12, ffn_dim=3072, layers=12, drop_path_rate=0.1)
    elif vit_type == "large":  # Q4: rop_path_rate typo -> drop path 0 (beit3.py:54)
        enc = dict(embed_dim=1024, heads=16, ffn_dim=4096, layers=24, drop_path_rate=0.0)
    elif vit_type == "tiny":   # test-only geometry (G2 fixtures), not a reference config
        enc = dict(embed_dim=128, heads=2, ffn_dim=256, layers=2, drop_path_rate=0.0)
    else:
        raise TypeError("please select the <vit_type> from ['base','large']")
    cfg = dict(vit_type=vit_type, img_size=img_size, patch_size=patch_size, max_token=max_token,
               vocab_size=64010, ln_eps=1e-5, num_queries=num_queries, head_dim=256, head_heads=8,
               dec_layers=3, dec_ffn=2048, tgqg_layers=2, tgqg_ffn=512, num_classes=1,
               eos_coef=0.1, cost_class=1.0, cost_bbox=5.0, cost_giou=2.0,
               w_class=1.0, w_bbox=5.0, w_giou=2.0, w_decoder=1.0, w_token=2.0, w_distill=1.0,
               branches=("decoder", "balanced_distill"), **enc)
    cfg.update(over)
    return SimpleNamespace(**cfg)


def cfg_from_branch_loss_weight(cfg, branch_loss_weight):
    """head kwarg `branch_loss_weight` (e.g. {"decoder": 1.0} of the *_twostage_1 / pre-training configs, or ViT-L's
    {"decoder": 1.0, "balanced_distill": {"token": 1.0, "distill": 0.4}}) -> the restatement's cfg fields."""
    if branch_loss_weight is None:
        return cfg
    assert set(branch_loss_weight) <= {"decoder", "balanced_distill"} and branch_loss_weight
    cfg.branches = tuple(k for k in ("decoder", "balanced_distill") if k in branch_loss_weight)
    if "decoder" in branch_loss_weight:
        cfg.w_decoder = float(branch_loss_weight["decoder"])
    if "balanced_distill" in branch_loss_weight:
        cfg.w_token = float(branch_loss_weight["balanced_distill"]["token"])
        cfg.w_distill = float(branch_loss_weight["balanced_distill"]["distill"])
    return cfg


# ----------------------------------------------------------------------------------------
# BEiT-3 multiway encoder  (vis_encs/beit/beit3_base.py:127-172,317-407,441-488 +
# torchscale leaf semantics)
# ----------------------------------------------------------------------------------------
def _mw(x, split, fa, fb):
    """torchscale MultiwayNetwork.forward: split at `split` on dim 1, A(x1) | B(x2)."""
    return torch.cat([fa(x[:, :split]), fb(x[:, split:])], dim=1)


def _mw_linear(sd, key, x, split):
    return _mw(x, split,
               lambda t: F.linear(t, sd[key + ".A.weight"], sd[key + ".A.bias"]),
               lambda t: F.linear(t, sd[key + ".B.weight"], sd[key + ".B.bias"]))


def _mw_ln(sd, key, x, split, eps):
    return _mw(x, split,
               lambda t: F.layer_norm(t, t.shape[-1:], sd[key + ".A.weight"], sd[key + ".A.bias"], eps),
               lambda t: F.layer_norm(t, t.shape[-1:], sd[key + ".B.weight"], sd[key + ".B.bias"], eps))


def encoder_embed(sd, cfg, img, ids, pad, p="vis_enc.beit3."):
    """beit3_base.py:461-475 (vision_embed + text_embed + cat), :317-334 (positions), :367 (pad zero)."""
    P = cfg.patch_size
    x1 = F.conv2d(img, sd[p + "vision_embed.proj.weight"], sd[p + "vision_embed.proj.bias"], stride=P)
    x1 = x1.flatten(2).transpose(1, 2)
    B = x1.shape[0]
    x1 = torch.cat([sd[p + "vision_embed.cls_token"].expand(B, -1, -1), x1], dim=1)
    split = x1.shape[1]
    x2 = F.embedding(ids, sd[p + "text_embed.weight"])
    x = torch.cat([x1, x2], dim=1)
    T = x2.shape[1]
    if pad is not None:
        mask = torch.cat([torch.zeros(x1.shape[:-1]).bool(), pad], dim=1)
    else:
        mask = torch.zeros(x.shape[:2]).bool()
    # PositionalEmbedding: indices start at 2 (fairseq convention)
    pos = torch.cat([sd[p + "encoder.embed_positions.A.weight"][2:2 + split],
                     sd[p + "encoder.embed_positions.B.weight"][2:2 + T]], dim=0)[None]
    x = x + pos  # embed_scale == 1.0 (no_scale_embedding)
    x = x * (1 - mask.unsqueeze(-1).type_as(x))
    return x, mask, split


def encoder_layer(sd, cfg, x, mask, split, i, dp_scale=None, p="vis_enc.beit3."):
    """beit3_base.py:127-172 with torchscale MultiheadAttention / FeedForwardNetwork (subln)."""
    L = f"{p}encoder.layers.{i}."
    B, N, D = x.shape
    H = cfg.heads
    d = D // H
    eps = cfg.ln_eps
    residual = x
    h = _mw_ln(sd, L + "self_attn_layer_norm", x, split, eps)
    q = _mw_linear(sd, L + "self_attn.q_proj", h, split) * (d ** -0.5)
    k = _mw_linear(sd, L + "self_attn.k_proj", h, split)
    v = _mw_linear(sd, L + "self_attn.v_proj", h, split)
    q = q.view(B, N, H, d).transpose(1, 2)
    k = k.view(B, N, H, d).transpose(1, 2)
    v = v.view(B, N, H, d).transpose(1, 2)
    w = q @ k.transpose(-1, -2)
    w = w.masked_fill(mask[:, None, None, :].to(torch.bool), float("-inf"))
    w = F.softmax(w, dim=-1, dtype=torch.float32)
    a = (w @ v).transpose(1, 2).reshape(B, N, D)
    a = _mw_ln(sd, L + "self_attn.inner_attn_ln", a, split, eps)
    a = _mw_linear(sd, L + "self_attn.out_proj", a, split)
    if dp_scale is not None and dp_scale[0] is not None:   # DropPath, attention branch (beit3_base.py:148-149)
        a = a * dp_scale[0][:, None, None]
    x = residual + a
    residual = x
    h = _mw_ln(sd, L + "final_layer_norm", x, split, eps)

    def ffn(t, e):
        t = F.linear(t, sd[f"{L}ffn.{e}.fc1.weight"], sd[f"{L}ffn.{e}.fc1.bias"])
        t = F.gelu(t.float()).type_as(t)
        t = F.layer_norm(t, t.shape[-1:], sd[f"{L}ffn.{e}.ffn_layernorm.weight"],
                         sd[f"{L}ffn.{e}.ffn_layernorm.bias"], eps)
        return F.linear(t, sd[f"{L}ffn.{e}.fc2.weight"], sd[f"{L}ffn.{e}.fc2.bias"])

    h = _mw(h, split, lambda t: ffn(t, "A"), lambda t: ffn(t, "B"))
    if dp_scale is not None and dp_scale[1] is not None:   # second, independent DropPath draw (:166-167)
        h = h * dp_scale[1][:, None, None]
    return residual + h


def beit3_forward(sd, cfg, img, ids, pad, dp_scales=None, p="vis_enc.beit3.", return_hidden=False):
    """BEIT3.forward (beit3.py:176-185): -> img_feat [B,HW,D], text_feat [B,T,D], cls_feat [B,D].

    dp_scales: optional list over layers of (attn_scale [B] | None, ffn_scale [B] | None) per-sample
    DropPath factors (bernoulli mask / keep_prob); None = eval mode.
    """
    x, mask, split = encoder_embed(sd, cfg, img, ids, pad, p)
    hidden = [x]
    for i in range(cfg.layers):
        x = encoder_layer(sd, cfg, x, mask, split, i, None if dp_scales is None else dp_scales[i], p)
        hidden.append(x)
    x = _mw_ln(sd, p + "encoder.layer_norm", x, split, cfg.ln_eps)
    T = ids.shape[-1]
    out = (x[:, 1:-T], x[:, -T:], x[:, 0])
    return (out, hidden) if return_hidden else out


# ----------------------------------------------------------------------------------------
# head: projections, TGQG, token branch, decoder branch
# (heads/tgqs_kd_detr_head/tgqs_kd_detr_head.py:375-454, transformer.py:134-235,
#  heads/utils.py:7-100; detrex leaf semantics SURVEY Appendix A.2)
# ----------------------------------------------------------------------------------------
def _mha(sd, key, query, keyt, value, heads, key_padding_mask=None, attn_drop_mask=None):
    """torch.nn.MultiheadAttention forward (batch-first here; packed in_proj q|k|v)."""
    W, bias = sd[key + ".attn.in_proj_weight"], sd[key + ".attn.in_proj_bias"]
    E = query.shape[-1]
    q = F.linear(query, W[:E], bias[:E])
    k = F.linear(keyt, W[E:2 * E], bias[E:2 * E])
    v = F.linear(value, W[2 * E:], bias[2 * E:])
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    d = E // heads
    q = q.view(B, Lq, heads, d).transpose(1, 2) * (d ** -0.5)
    k = k.view(B, Lk, heads, d).transpose(1, 2)
    v = v.view(B, Lk, heads, d).transpose(1, 2)
    w = q @ k.transpose(-1, -2)
    if key_padding_mask is not None:
        w = w.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    w = F.softmax(w, dim=-1)
    if attn_drop_mask is not None:
        w = w * attn_drop_mask
    o = (w @ v).transpose(1, 2).reshape(B, Lq, E)
    return F.linear(o, sd[key + ".attn.out_proj.weight"], sd[key + ".attn.out_proj.bias"])


def _ln(sd, key, x, eps=1e-5):
    return F.layer_norm(x, x.shape[-1:], sd[key + ".weight"], sd[key + ".bias"], eps)


def decoder_stack(sd, prefix, n_layers, heads, tgt, memory, query_pos, key_pos, key_padding_mask,
                  return_intermediate):
    """DetrTransformerDecoder.forward (transformer.py:134-186), post-norm layers
    ("self_attn","norm","cross_attn","norm","ffn","norm"), eval-mode (dropout identity)."""
    q = tgt
    inter = []
    for i in range(n_layers):
        L = f"{prefix}layers.{i}."
        qp = q + query_pos
        q = _ln(sd, L + "norms.0", q + _mha(sd, L + "attentions.0", qp, qp, q, heads))
        q = _ln(sd, L + "norms.1", q + _mha(sd, L + "attentions.1", q + query_pos, memory + key_pos,
                                            memory, heads, key_padding_mask))
        f = F.linear(F.relu(F.linear(q, sd[L + "ffns.0.layers.0.0.weight"], sd[L + "ffns.0.layers.0.0.bias"])),
                     sd[L + "ffns.0.layers.1.weight"], sd[L + "ffns.0.layers.1.bias"])
        q = _ln(sd, L + "norms.2", q + f)
        if return_intermediate:
            inter.append(_ln(sd, prefix + "post_norm_layer", q))
    if return_intermediate:
        return torch.stack(inter)
    return _ln(sd, prefix + "post_norm_layer", q)[None]


def sine_pos_2d(mask, num_pos_feats=128, temperature=10000, scale=2 * math.pi, eps=1e-6):
    """detrex PositionEmbeddingSine(normalize=True) -> [B, 2*num_pos_feats, H, W]."""
    not_mask = ~mask
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    B, H, W = mask.shape
    pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=4).view(B, H, W, -1)
    pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=4).view(B, H, W, -1)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def sine_pos_1d(pos_len, dim):
    """PositionEmbeddingSine1D.forward (heads/utils.py:72-100), quirk Q2: the frequency
    vector is cast to long -> [1,0,0,...]."""
    i_matrix = torch.arange(dim // 2, dtype=torch.float)
    i_matrix /= dim / 2
    i_matrix = (1 / torch.pow(10000, i_matrix)).to(torch.long)
    out = torch.arange(pos_len).to(torch.long)[:, None] @ i_matrix[None, :]
    emb = torch.zeros(pos_len, dim, dtype=torch.float)
    emb[:, 0::2] = torch.sin(out)
    emb[:, 1::2] = torch.cos(out)
    return emb


def image_masks(B, feat_hw, img_metas):
    """x_mask_pos_enc (tgqs_kd_detr_head.py:322-338): padding mask at feature resolution."""
    try:
        Hin, Win = img_metas[0]["batch_input_shape"]
    except Exception:
        Hin, Win, _ = img_metas[0]["img_shape"]
    m = torch.ones((B, Hin, Win))
    for i in range(B):
        h, w, _ = img_metas[i]["img_shape"]
        m[i, :h, :w] = 0
    return F.interpolate(m.unsqueeze(1), size=feat_hw).to(torch.bool).squeeze(1)


def _mlp(sd, key, x, n):
    for i in range(n):
        x = F.linear(x, sd[f"{key}.layers.{i}.weight"], sd[f"{key}.layers.{i}.bias"])
        if i < n - 1:
            x = F.relu(x)
    return x


def head_forward_general(sd, cfg, img_feat, text_feat, cls_feat, text_mask, img_metas, p="head."):
    """forward_general (tgqs_kd_detr_head.py:375-454), eval-mode dropout.
    img_feat: [B, HW, C] token-major (the reference reshapes to [B,C,h,w] then flattens back)."""
    B, HW, C = img_feat.shape
    E, nq = cfg.head_dim, cfg.num_queries
    hw = int(round(HW ** 0.5))
    mem = F.linear(img_feat, sd[p + "input_proj.weight"].view(E, C), sd[p + "input_proj.bias"])  # 1x1 conv
    text = F.linear(text_feat, sd[p + "input_text_proj.weight"], sd[p + "input_text_proj.bias"])
    cls = F.linear(cls_feat, sd[p + "input_cls_proj.weight"], sd[p + "input_cls_proj.bias"]).unsqueeze(1)
    masks = image_masks(B, (hw, hw), img_metas)
    pos = sine_pos_2d(masks, E // 2).flatten(2).transpose(1, 2)           # [B, HW, E]
    cls = cls.repeat(1, nq, 1)
    # ---- TGQG (:385-399).  Q1: `~text_mask` on an int64 mask is bitwise NOT -> integer indexing.
    filt = torch.cat([torch.max(f[m, :], dim=0, keepdim=True)[0] for f, m in zip(text, ~text_mask)])
    filt = filt.unsqueeze(1).repeat(1, nq, 1)
    qe = sd[p + "query_embed.weight"].unsqueeze(0).repeat(B, 1, 1)         # [B, nq, E]
    tpos = sine_pos_1d(text.shape[1], E)[None]
    g = decoder_stack(sd, p + "text_guided_query_generation_transformer.", cfg.tgqg_layers, cfg.head_heads,
                      torch.zeros_like(qe), text, qe, tpos, text_mask.bool(), return_intermediate=False)
    query_embed = g[0] + filt + qe
    tok = query_embed + cls                                               # Q5
    # ---- token branch (:403-420), num_token_mlp_layers=1, return_intermediate=True; skipped -- None outputs, in
    # forward_test too -- when branch_loss_weight is {"decoder": w} alone (:403-409)
    if tuple(cfg.branches) == ("decoder",):
        tok_logits = tok_boxes = None
    else:
        tok = _mlp(sd, p + "mlp", tok, 1)[None]                           # [1,B,nq,E]
        tok_logits = F.linear(tok, sd[p + "class_embed_token.weight"], sd[p + "class_embed_token.bias"])
        tok_boxes = _mlp(sd, p + "bbox_embed_token", tok, 3).sigmoid()
    # ---- decoder branch (:425-428)
    hs = decoder_stack(sd, p + "transformer.decoder.", cfg.dec_layers, cfg.head_heads,
                       torch.zeros_like(query_embed), mem, query_embed, pos, masks.flatten(1),
                       return_intermediate=True)                           # [L,B,nq,E]
    dec_logits = F.linear(hs, sd[p + "class_embed_decoder.weight"], sd[p + "class_embed_decoder.bias"])
    dec_boxes = _mlp(sd, p + "bbox_embed_decoder", hs, 3).sigmoid()
    return dict(tok_logits=tok_logits, tok_boxes=tok_boxes, dec_logits=dec_logits, dec_boxes=dec_boxes,
                token_features=tok, decoder_features=hs, query_embed=query_embed)


# ----------------------------------------------------------------------------------------
# matcher + criterion (detrex HungarianMatcher ce_cost; core/criterion/criterion.py:108-271)
# ----------------------------------------------------------------------------------------
def matcher_cost(logits, boxes, tgt_ids, tgt_boxes, cfg):
    prob = logits.flatten(0, 1).softmax(-1)
    ob = boxes.flatten(0, 1)
    return (cfg.cost_bbox * torch.cdist(ob, tgt_boxes, p=1) + cfg.cost_class * (-prob[:, tgt_ids])
            + cfg.cost_giou * (-generalized_box_iou(box_cxcywh_to_xyxy(ob), box_cxcywh_to_xyxy(tgt_boxes))))


@torch.no_grad()
def hungarian(logits, boxes, targets, cfg):
    from scipy.optimize import linear_sum_assignment
    B, nq = logits.shape[:2]
    tgt_ids = torch.cat([t["labels"] for t in targets])
    tgt_boxes = torch.cat([t["boxes"] for t in targets])
    C = matcher_cost(logits, boxes, tgt_ids, tgt_boxes, cfg).view(B, nq, -1)
    sizes = [len(t["boxes"]) for t in targets]
    out = []
    for i, c in enumerate(C.split(sizes, -1)):
        r, cidx = linear_sum_assignment(c[i])
        out.append((torch.as_tensor(r, dtype=torch.int64), torch.as_tensor(cidx, dtype=torch.int64)))
    return out


def _criterion_one(logits, boxes, targets, indices, num_boxes, cfg):
    """loss_labels (ce_loss) + loss_boxes (criterion.py:108-127,164-204)."""
    B, nq = logits.shape[:2]
    bidx = torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(indices)])
    sidx = torch.cat([s for s, _ in indices])
    tco = torch.cat([t["labels"][J] for t, (_, J) in zip(targets, indices)])
    tc = torch.full((B, nq), cfg.num_classes, dtype=torch.int64)
    tc[(bidx, sidx)] = tco
    ew = torch.ones(cfg.num_classes + 1)
    ew[-1] = cfg.eos_coef
    loss_class = F.cross_entropy(logits.transpose(1, 2), tc, ew)
    src = boxes[(bidx, sidx)]
    tb = torch.cat([t["boxes"][J] for t, (_, J) in zip(targets, indices)], dim=0)
    loss_bbox = F.l1_loss(src, tb, reduction="none").sum() / num_boxes
    loss_giou = (1 - torch.diag(generalized_box_iou(box_cxcywh_to_xyxy(src), box_cxcywh_to_xyxy(tb)))).sum() / num_boxes
    return loss_class, loss_bbox, loss_giou


def set_criterion(all_logits, all_boxes, targets, cfg, world_size=1, return_indices=False):
    """SetCriterion.forward + calc_loss weighting (criterion.py:226-271, tgqs_kd_detr_head.py:340-350).
    all_logits/all_boxes: [L,B,nq,*]; last = main output, the rest = aux_outputs. Returns the
    WEIGHTED loss dict (Q6: calc_loss multiplies in place by weight_dict incl. _i aux keys)."""
    num_boxes = float(sum(len(t["labels"]) for t in targets))
    num_boxes = max(num_boxes / world_size, 1.0)                            # Q7
    losses, all_idx = {}, []
    order = [(all_logits.shape[0] - 1, "")] + [(i, f"_{i}") for i in range(all_logits.shape[0] - 1)]
    for l, suf in order:
        idx = hungarian(all_logits[l].detach(), all_boxes[l].detach(), targets, cfg)
        all_idx.append(idx)
        lc, lb, lg = _criterion_one(all_logits[l], all_boxes[l], targets, idx, num_boxes, cfg)
        losses["loss_class" + suf] = lc * cfg.w_class
        losses["loss_bbox" + suf] = lb * cfg.w_bbox
        losses["loss_giou" + suf] = lg * cfg.w_giou
    return (losses, all_idx) if return_indices else losses


def prepare_soft_targets(gt_bbox, dec_logits, dec_boxes, img_metas, cfg):
    """prepare_soft_targets, mode score_iou_weighted (tgqs_kd_detr_head.py:207-268)."""
    logits, boxes = dec_logits.detach(), dec_boxes.detach()
    scores = F.softmax(logits, dim=-1)[:, :, 0:1]
    tg = []
    for tb, meta in zip(gt_bbox, img_metas):
        h, w = meta["img_shape"][:2]
        whwh = torch.as_tensor([w, h, w, h], dtype=torch.float)
        if tb.dim() == 1:
            tb_ = tb.unsqueeze(0)
            cls = torch.zeros(1).long()
        else:  # GRefCOCO: drop no-target entries
            assert int(tb.shape[0]) == len(meta["target"])
            keep = [i for i, t in enumerate(meta["target"]) if t["category_id"] != -1]
            tb_ = tb[keep] if len(keep) else torch.zeros((0, 4))
            cls = torch.zeros(len(keep)).long()
        tg.append({"labels": cls, "boxes": box_xyxy_to_cxcywh(tb_.float() / whwh).float()})
    idx = hungarian(logits, boxes, tg, cfg)
    tp = []
    for (i0, i1), pb, ps, t in zip(idx, boxes, scores, tg):
        pb_ = pb[i0]
        ious = torch.diag(box_iou(box_cxcywh_to_xyxy(pb_), box_cxcywh_to_xyxy(t["boxes"][i1]))[0])
        wgt = ps[i0].reshape(-1) * ious
        tp.append({"labels": torch.zeros(pb_.shape[0]).long(), "boxes": pb_, "weight": wgt})
    return tg, tp


def head_forward_train(sd, cfg, out, gt_bbox, img_metas, world_size=1):
    """forward_train (tgqs_kd_detr_head.py:456-572): the "decoder" (:483-487) and "balanced_distill" (:489-509) blocks
    are independent `if`s over branch_loss_weight's keys (cfg.branches); a block that is absent leaves its term at the
    constant 0 of :474-480 and adds no entry to the loss dict; loss_total = the sum (:571)."""
    tg, tp = prepare_soft_targets(gt_bbox, out["dec_logits"][-1], out["dec_boxes"][-1], img_metas, cfg)
    losses, detail = {}, dict(targets_gt=tg, targets_pred=tp)
    total = torch.tensor(0.0)
    if "decoder" in cfg.branches:
        ld = set_criterion(out["dec_logits"], out["dec_boxes"], tg, cfg, world_size)
        losses["loss_dgt"] = cfg.w_decoder * sum(ld.values())
        total = total + losses["loss_dgt"]
        detail["dec"] = ld
    if "balanced_distill" in cfg.branches:
        w = torch.mean(torch.cat([t["weight"] for t in tp]))                # Q8 (detached)
        tl, tbx = out["tok_logits"][-1:], out["tok_boxes"][-1:]
        lt = set_criterion(tl, tbx, tg, cfg, world_size)
        losses["loss_tgt"] = cfg.w_token * sum(lt.values()) * (1 - w)
        lk = set_criterion(tl, tbx, tp, cfg, world_size)
        losses["loss_kd"] = cfg.w_distill * sum(lk.values()) * w
        losses["loss_distill_w"] = w
        total = total + losses["loss_tgt"] + losses["loss_kd"]
        detail.update(tok_gt=lt, tok_kd=lk)
    losses["loss_total"] = total
    return losses, detail


# ----------------------------------------------------------------------------------------
# post-processing (head.inference :577-604, MIXDETRMB.get_predictions[_grec]
# det_seg/mix_detr_mb.py:127-190, detectron2 leaf SURVEY A.4)
# ----------------------------------------------------------------------------------------
def get_predictions(logits, boxes, img_metas, rescale=False, grec=False):
    scores, labels = F.softmax(logits, dim=-1)[:, :, :-1].max(-1)
    pb, pc, res = [], [], []
    for s, l, b, meta in zip(scores, labels, boxes, img_metas):
        h, w = meta["img_shape"][:2]
        xy = box_cxcywh_to_xyxy(b).clone()
        xy[:, 0::2] *= w
        xy[:, 1::2] *= h
        xy = torch.stack((xy[:, 0].clamp(0, w), xy[:, 1].clamp(0, h), xy[:, 2].clamp(0, w), xy[:, 3].clamp(0, h)), -1)
        keep = ((xy[:, 2] - xy[:, 0]) > 0) & ((xy[:, 3] - xy[:, 1]) > 0)
        xy, s, l = xy[keep], s[keep], l[keep]
        if grec:
            if rescale:
                xy = xy / xy.new_tensor(meta["scale_factor"])
            res.append({"boxes": xy, "scores": s, "labels": l})
            continue
        best = int(torch.argmax(s))
        box = xy[best].view(1, -1)
        if rescale:
            box = box / box.new_tensor(meta["scale_factor"])
        pb.append(box)
        pc.append(l)
    if grec:
        return dict(pred_bboxes=res, pred_masks=None)
    return dict(pred_bboxes=torch.cat(pb, 0), pred_masks=None, predict_classes=torch.cat(pc, 0))


# ----------------------------------------------------------------------------------------
# whole-model entry points (det_seg/mix_detr_mb.py:19-125, det_seg/base.py:12-27)
# ----------------------------------------------------------------------------------------
def model_forward(sd, cfg, img, ids, img_metas, text_attention_mask, dp_scales=None):
    for m in img_metas:
        m["batch_input_shape"] = tuple(img.shape[-2:])
    img_feat, text_feat, cls_feat = beit3_forward(sd, cfg, img, ids, text_attention_mask, dp_scales)
    return head_forward_general(sd, cfg, img_feat, text_feat, cls_feat, text_attention_mask, img_metas)


@torch.no_grad()
def forward_test(sd, cfg, img, ids, img_metas, text_attention_mask, rescale=False):
    out = model_forward(sd, cfg, img, ids, img_metas, text_attention_mask)
    grec = img_metas[0].get("target", None) is not None
    if out["tok_logits"] is None:          # get_predictions on a {"pred_logits": None} output (mix_detr_mb.py:128-129,162-163)
        pt = dict(pred_bboxes=None, pred_masks=None, predict_classes=None)
    else:
        pt = get_predictions(out["tok_logits"][-1], out["tok_boxes"][-1], img_metas, rescale, grec)
    pd = get_predictions(out["dec_logits"][-1], out["dec_boxes"][-1], img_metas, rescale, grec)
    return [pd, pt], out


def forward_train(sd, cfg, img, ids, img_metas, text_attention_mask, gt_bbox, dp_scales=None, world_size=1):
    out = model_forward(sd, cfg, img, ids, img_metas, text_attention_mask, dp_scales)
    losses, detail = head_forward_train(sd, cfg, out, gt_bbox, img_metas, world_size)
    return losses, out, detail
