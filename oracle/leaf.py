"""Third-party LEAF semantics restated for the oracle (TEST INFRASTRUCTURE ONLY).

This file is part of ``oracle/`` -- the CPU checker for the HIP hot path. Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it; the product package ``simvg_amd`` never does.

The reference (``/root/reference``) imports its encoder / decoder arithmetic from
packages that are NOT vendored in the reference tree and are not installed here:

* ``torchscale``  (unpinned at reference ``requirements.txt:13``; BEiT-3 upstream pins 0.2.0;
  call sites ``simvg/models/vis_encs/beit/beit3_base.py:8-32``)
* ``detrex``      (git HEAD, reference ``README.md:57-65``; call sites
  ``simvg/models/heads/tgqs_kd_detr_head/transformer.py:19``,
  ``tgqs_kd_detr_head.py:7-10``, ``simvg/core/criterion/criterion.py:27-28``)
* ``detectron2``  (git HEAD; call sites ``simvg/models/det_seg/mix_detr_mb.py:7,10``)
* ``mmcv`` 1.7.2  (``simvg/models/builder.py:1``, ``simvg/models/det_seg/base.py:2``)

The classes below restate the PUBLISHED semantics of exactly the symbols those call
sites use.  PARITY UNPINNED at this boundary: the reference holds
no tests / golden vectors, and the real packages cannot be imported here, so these leaf
classes are anchored on upstream's documented behaviour and on ``torch.nn`` primitives
(``nn.Linear``, ``nn.LayerNorm``, ``nn.MultiheadAttention``, ``F.gelu`` ...) which ARE
present.  Everything authored inside the reference tree is pinned by executing the
reference files themselves on top of these leaves (``oracle/ref_loader.py``).
"""
import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# torchscale 0.2.0
# --------------------------------------------------------------------------------------
class EncoderConfig:
    """torchscale.architecture.config.EncoderConfig -- kwargs bag with defaults.

    Unknown kwargs are silently ignored, which is how the reference's ``rop_path_rate``
    typo (``beit3.py:54``) passes.
    """

    def __init__(self, **kw):
        g = kw.pop
        self.encoder_embed_dim = g("encoder_embed_dim", 768)
        self.encoder_attention_heads = g("encoder_attention_heads", 12)
        self.encoder_ffn_embed_dim = g("encoder_ffn_embed_dim", 3072)
        self.encoder_layers = g("encoder_layers", 12)
        self.encoder_normalize_before = g("encoder_normalize_before", True)
        self.normalize_output = g("normalize_output", True)
        self.activation_fn = g("activation_fn", "gelu")
        self.dropout = g("dropout", 0.0)
        self.drop_path_rate = g("drop_path_rate", 0.0)
        self.attention_dropout = g("attention_dropout", 0.0)
        self.activation_dropout = g("activation_dropout", 0.0)
        self.no_scale_embedding = g("no_scale_embedding", True)
        self.layernorm_embedding = g("layernorm_embedding", False)
        self.moe_freq = g("moe_freq", 0)
        self.moe_top1_expert = g("moe_top1_expert", False)
        self.moe_expert_count = g("moe_expert_count", 0)
        self.rel_pos_buckets = g("rel_pos_buckets", 0)
        self.max_rel_pos = g("max_rel_pos", 0)
        self.deepnorm = g("deepnorm", False)
        self.subln = g("subln", True)
        self.bert_init = g("bert_init", False)
        self.multiway = g("multiway", False)
        self.share_encoder_input_output_embed = g("share_encoder_input_output_embed", False)
        self.max_source_positions = g("max_source_positions", 1024)
        self.no_output_layer = g("no_output_layer", False)
        self.layernorm_eps = g("layernorm_eps", 1e-5)
        self.vocab_size = g("vocab_size", -1)
        self.img_size = g("img_size", 224)
        self.patch_size = g("patch_size", 16)
        self.in_chans = g("in_chans", 3)
        self.checkpoint_activations = g("checkpoint_activations", False)
        self.fsdp = g("fsdp", False)
        self.ddp_rank = g("ddp_rank", 0)
        self.xpos_rel_pos = g("xpos_rel_pos", False)
        self.xpos_scale_base = g("xpos_scale_base", 512)
        if self.deepnorm:
            self.encoder_normalize_before = False
            self.subln = False
        if self.subln:
            self.encoder_normalize_before = True
            self.deepnorm = False


def init_bert_params(module):  # unused: bert_init=False
    pass


class MultiwayNetwork(nn.Module):
    def __init__(self, module, dim=1):
        super().__init__()
        self.dim = dim
        self.A = module
        self.B = copy.deepcopy(module)
        self.B.reset_parameters()
        self.split_position = -1

    def forward(self, x, **kwargs):
        if self.split_position == -1:
            return self.A(x, **kwargs)
        if self.split_position == 0:
            return self.B(x, **kwargs)
        x1, x2 = torch.split(
            x, [self.split_position, x.size(self.dim) - self.split_position], dim=self.dim
        )
        y1, y2 = self.A(x1, **kwargs), self.B(x2, **kwargs)
        return torch.cat([y1, y2], dim=self.dim)


def MultiwayWrapper(args, module, dim=1):
    if args.multiway:
        return MultiwayNetwork(module, dim=dim)
    return module


class MutliwayEmbedding(MultiwayNetwork):  # (sic) upstream spelling
    def __init__(self, modules, dim=1):
        super(MultiwayNetwork, self).__init__()
        self.dim = dim
        assert len(modules) == 2
        self.A = modules[0]
        self.B = modules[1]
        self.split_position = -1


def set_split_position(position):
    def apply_fn(module):
        if hasattr(module, "split_position"):
            module.split_position = position

    return apply_fn


class VisionEmbedding(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768,
                 contain_mask_token=False, prepend_cls_token=False):
        super().__init__()
        img_size = (img_size, img_size)
        patch_size = (patch_size, patch_size)
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0])
        self.patch_shape = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.img_size = img_size
        self.patch_size = patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if contain_mask_token else None
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if prepend_cls_token else None

    def num_position_embeddings(self):
        return self.num_patches if self.cls_token is None else self.num_patches + 1

    def forward(self, x, masked_position=None, **kwargs):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        x = self.proj(x).flatten(2).transpose(1, 2)
        batch_size, seq_len, _ = x.size()
        if masked_position is not None:
            assert self.mask_token is not None
            mask_token = self.mask_token.expand(batch_size, seq_len, -1)
            w = masked_position.unsqueeze(-1).type_as(mask_token)
            x = x * (1 - w) + mask_token * w
        if self.cls_token is not None:
            cls_tokens = self.cls_token.expand(batch_size, -1, -1)
            x = torch.cat((cls_tokens, x), dim=1)
        return x


class TextEmbedding(nn.Embedding):
    def reset_parameters(self):
        nn.init.normal_(self.weight, mean=0, std=self.embedding_dim ** -0.5)
        self._fill_padding_idx_with_zero()


class PositionalEmbedding(nn.Embedding):
    def forward(self, x, positions=None, **kwargs):
        if positions is None:
            # being consistent with Fairseq, which starts from 2.
            positions = torch.arange(2, x.size(1) + 2, device=x.device).long().unsqueeze(0)
        return F.embedding(positions, self.weight, self.padding_idx, self.max_norm,
                           self.norm_type, self.scale_grad_by_freq, self.sparse)


class DropPath(nn.Module):
    """timm stochastic depth (per-sample), as wrapped by torchscale.component.droppath."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep_prob = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        random_tensor = x.new_empty(shape).bernoulli_(keep_prob)
        if keep_prob > 0.0:
            random_tensor.div_(keep_prob)
        return x * random_tensor


def _get_activation_fn(activation):
    if activation == "relu":
        return F.relu
    if activation == "gelu":
        return F.gelu
    raise NotImplementedError


class FeedForwardNetwork(nn.Module):
    def __init__(self, embed_dim, ffn_dim, activation_fn, dropout, activation_dropout,
                 layernorm_eps, subln=False):
        super().__init__()
        self.embed_dim = embed_dim
        self.activation_fn = _get_activation_fn(activation=str(activation_fn))
        self.activation_dropout_module = torch.nn.Dropout(activation_dropout)
        self.dropout_module = torch.nn.Dropout(dropout)
        self.fc1 = nn.Linear(self.embed_dim, ffn_dim)
        self.fc2 = nn.Linear(ffn_dim, self.embed_dim)
        self.ffn_layernorm = nn.LayerNorm(ffn_dim, eps=layernorm_eps) if subln else None

    def reset_parameters(self):
        self.fc1.reset_parameters()
        self.fc2.reset_parameters()
        if self.ffn_layernorm is not None:
            self.ffn_layernorm.reset_parameters()

    def forward(self, x):
        x_shape = x.shape
        x = x.reshape(-1, x.size(-1))
        x = self.fc1(x)
        x = self.activation_fn(x.float()).type_as(x)
        x = self.activation_dropout_module(x)
        if self.ffn_layernorm is not None:
            x = self.ffn_layernorm(x)
        x = self.fc2(x)
        x = x.view(x_shape)
        x = self.dropout_module(x)
        return x


def make_experts(*a, **k):  # MoE unused (moe_freq=0)
    raise NotImplementedError


class TSMultiheadAttention(nn.Module):
    """torchscale.component.multihead_attention.MultiheadAttention (<=0.2.0: 2-tuple return)."""

    def __init__(self, args, embed_dim, num_heads, dropout=0.0, self_attention=False,
                 encoder_decoder_attention=False, subln=False):
        super().__init__()
        self.args = args
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.head_dim = embed_dim // num_heads
        self.scaling = self.head_dim ** -0.5
        self.self_attention = self_attention
        self.encoder_decoder_attention = encoder_decoder_attention
        assert self.self_attention ^ self.encoder_decoder_attention
        self.k_proj = MultiwayWrapper(args, nn.Linear(embed_dim, embed_dim, bias=True))
        self.v_proj = MultiwayWrapper(args, nn.Linear(embed_dim, embed_dim, bias=True))
        self.q_proj = MultiwayWrapper(args, nn.Linear(embed_dim, embed_dim, bias=True))
        self.out_proj = MultiwayWrapper(args, nn.Linear(embed_dim, embed_dim, bias=True))
        self.inner_attn_ln = (
            MultiwayWrapper(args, nn.LayerNorm(self.embed_dim, eps=args.layernorm_eps))
            if subln and self.self_attention else None
        )
        self.dropout_module = torch.nn.Dropout(dropout)
        self.xpos = None

    def forward(self, query, key, value, incremental_state=None, key_padding_mask=None,
                attn_mask=None, rel_pos=None):
        bsz, tgt_len, embed_dim = query.size()
        src_len = tgt_len
        assert embed_dim == self.embed_dim
        key_bsz, src_len, _ = key.size()
        q = self.q_proj(query)
        k = self.k_proj(key)
        v = self.v_proj(value)
        q = q * self.scaling
        q = q.view(bsz, tgt_len, self.num_heads, self.head_dim).transpose(1, 2)
        k = k.view(bsz, src_len, self.num_heads, self.head_dim).transpose(1, 2)
        v = v.view(bsz, src_len, self.num_heads, self.head_dim).transpose(1, 2)
        q = q.reshape(bsz * self.num_heads, tgt_len, self.head_dim)
        k = k.reshape(bsz * self.num_heads, src_len, self.head_dim)
        v = v.reshape(bsz * self.num_heads, src_len, self.head_dim)
        attn_weights = torch.bmm(q, k.transpose(1, 2))
        if attn_mask is not None:
            attn_weights = torch.nan_to_num(attn_weights)
            attn_mask = attn_mask.unsqueeze(0)
            attn_weights += attn_mask
        if key_padding_mask is not None:
            attn_weights = attn_weights.view(bsz, self.num_heads, tgt_len, src_len)
            attn_weights = attn_weights.masked_fill(
                key_padding_mask.unsqueeze(1).unsqueeze(2).to(torch.bool), float("-inf"))
            attn_weights = attn_weights.view(bsz * self.num_heads, tgt_len, src_len)
        attn_weights = F.softmax(attn_weights, dim=-1, dtype=torch.float32).type_as(attn_weights)
        attn_probs = self.dropout_module(attn_weights)
        attn = torch.bmm(attn_probs, v)
        attn = attn.transpose(0, 1).reshape(tgt_len, bsz, embed_dim).transpose(0, 1)
        if self.inner_attn_ln is not None:
            attn = self.inner_attn_ln(attn)
        attn = self.out_proj(attn)
        attn_weights = attn_weights.view(bsz, self.num_heads, tgt_len, src_len).transpose(1, 0)
        return attn, attn_weights


# --------------------------------------------------------------------------------------
# detrex layers
# --------------------------------------------------------------------------------------
class DxMultiheadAttention(nn.Module):
    """detrex.layers.MultiheadAttention: wrapper of torch.nn.MultiheadAttention with
    identity connection and positional encodings added to query / key (NOT value)."""

    def __init__(self, embed_dim, num_heads, attn_drop=0.0, proj_drop=0.0, batch_first=False, **kw):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.batch_first = batch_first
        self.attn = nn.MultiheadAttention(embed_dim=embed_dim, num_heads=num_heads,
                                          dropout=attn_drop, batch_first=batch_first, **kw)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None,
                attn_mask=None, key_padding_mask=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None:
            if query_pos is not None:
                if query_pos.shape == key.shape:
                    key_pos = query_pos
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        out = self.attn(query=query, key=key, value=value, attn_mask=attn_mask,
                        key_padding_mask=key_padding_mask)[0]
        return identity + self.proj_drop(out)


class FFN(nn.Module):
    def __init__(self, embed_dim=256, feedforward_dim=1024, output_dim=None, num_fcs=2,
                 activation=nn.ReLU(inplace=True), ffn_drop=0.0, fc_bias=True, add_identity=True):
        super().__init__()
        assert num_fcs >= 2
        self.embed_dim = embed_dim
        self.feedforward_dim = feedforward_dim
        self.num_fcs = num_fcs
        self.activation = activation
        output_dim = embed_dim if output_dim is None else output_dim
        layers = []
        in_channels = embed_dim
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(nn.Linear(in_channels, feedforward_dim, bias=fc_bias),
                                        self.activation, nn.Dropout(ffn_drop)))
            in_channels = feedforward_dim
        layers.append(nn.Linear(feedforward_dim, output_dim, bias=fc_bias))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = nn.Sequential(*layers)
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return out
        if identity is None:
            identity = x
        return identity + out


class BaseTransformerLayer(nn.Module):
    def __init__(self, attn, ffn, norm, operation_order=None):
        super().__init__()
        assert set(operation_order).issubset({"self_attn", "norm", "cross_attn", "ffn"})
        num_attn = operation_order.count("self_attn") + operation_order.count("cross_attn")
        if isinstance(attn, nn.Module):
            attn = [copy.deepcopy(attn) for _ in range(num_attn)]
        self.num_attn = num_attn
        self.operation_order = operation_order
        self.pre_norm = operation_order[0] == "norm"
        self.attentions = nn.ModuleList()
        index = 0
        for operation_name in operation_order:
            if operation_name in ["self_attn", "cross_attn"]:
                self.attentions.append(attn[index])
                index += 1
        self.embed_dim = self.attentions[0].embed_dim
        self.ffns = nn.ModuleList()
        num_ffns = operation_order.count("ffn")
        for _ in range(num_ffns):
            self.ffns.append(copy.deepcopy(ffn))
        self.norms = nn.ModuleList()
        num_norms = operation_order.count("norm")
        for _ in range(num_norms):
            self.norms.append(copy.deepcopy(norm))

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        norm_index = attn_index = ffn_index = 0
        identity = query
        if attn_masks is None:
            attn_masks = [None for _ in range(self.num_attn)]
        elif isinstance(attn_masks, torch.Tensor):
            attn_masks = [copy.deepcopy(attn_masks) for _ in range(self.num_attn)]
        for layer in self.operation_order:
            if layer == "self_attn":
                temp_key = temp_value = query
                query = self.attentions[attn_index](
                    query, temp_key, temp_value, identity if self.pre_norm else None,
                    query_pos=query_pos, key_pos=query_pos, attn_mask=attn_masks[attn_index],
                    key_padding_mask=query_key_padding_mask, **kwargs)
                attn_index += 1
                identity = query
            elif layer == "norm":
                query = self.norms[norm_index](query)
                norm_index += 1
            elif layer == "cross_attn":
                query = self.attentions[attn_index](
                    query, key, value, identity if self.pre_norm else None,
                    query_pos=query_pos, key_pos=key_pos, attn_mask=attn_masks[attn_index],
                    key_padding_mask=key_padding_mask, **kwargs)
                attn_index += 1
                identity = query
            elif layer == "ffn":
                query = self.ffns[ffn_index](query, identity if self.pre_norm else None)
                ffn_index += 1
        return query


class TransformerLayerSequence(nn.Module):
    def __init__(self, transformer_layers=None, num_layers=None):
        super().__init__()
        self.num_layers = num_layers
        self.layers = nn.ModuleList()
        if isinstance(transformer_layers, nn.Module):
            for _ in range(num_layers):
                self.layers.append(copy.deepcopy(transformer_layers))
        else:
            assert isinstance(transformer_layers, list) and len(transformer_layers) == num_layers
            for l in transformer_layers:
                self.layers.append(l)

    def forward(self):
        raise NotImplementedError()


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, scale=2 * math.pi, eps=1e-6,
                 offset=0.0, normalize=False):
        super().__init__()
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.normalize = normalize
        self.scale = scale
        self.eps = eps
        self.offset = offset

    def forward(self, mask, **kwargs):
        assert mask is not None
        not_mask = ~mask
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            y_embed = (y_embed + self.offset) / (y_embed[:, -1:, :] + self.eps) * self.scale
            x_embed = (x_embed + self.offset) / (x_embed[:, :, -1:] + self.eps) * self.scale
        dim_t = torch.arange(self.num_pos_feats, dtype=torch.float32, device=mask.device)
        dim_t = self.temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / self.num_pos_feats)
        pos_x = x_embed[:, :, :, None] / dim_t
        pos_y = y_embed[:, :, :, None] / dim_t
        B, H, W = mask.size()
        pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
        pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
        pos = torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)
        return pos


class PositionEmbeddingLearned(nn.Module):  # imported by the reference head, never built
    pass


def box_cxcywh_to_xyxy(bbox):
    cx, cy, w, h = bbox.unbind(-1)
    return torch.stack([(cx - 0.5 * w), (cy - 0.5 * h), (cx + 0.5 * w), (cy + 0.5 * h)], dim=-1)


def box_xyxy_to_cxcywh(bbox):
    x0, y0, x1, y1 = bbox.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, (x1 - x0), (y1 - y0)], dim=-1)


def _box_area(boxes):
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def box_iou(boxes1, boxes2):
    area1 = _box_area(boxes1)
    area2 = _box_area(boxes2)
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = area1[:, None] + area2 - inter
    iou = inter / (union + 1e-6)
    return iou, union


def generalized_box_iou(boxes1, boxes2):
    assert (boxes1[:, 2:] >= boxes1[:, :2]).all()
    assert (boxes2[:, 2:] >= boxes2[:, :2]).all()
    iou, union = box_iou(boxes1, boxes2)
    lt = torch.min(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.max(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    area = wh[:, :, 0] * wh[:, :, 1]
    return iou - (area - union) / (area + 1e-6)


class HungarianMatcher(nn.Module):
    """detrex.modeling.matcher.HungarianMatcher (ce_cost branch) + SciPy LSAP on the host."""

    def __init__(self, cost_class=1, cost_bbox=1, cost_giou=1, cost_class_type="focal_loss_cost",
                 alpha=0.25, gamma=2.0):
        super().__init__()
        self.cost_class = cost_class
        self.cost_bbox = cost_bbox
        self.cost_giou = cost_giou
        self.cost_class_type = cost_class_type
        self.alpha = alpha
        self.gamma = gamma
        assert cost_class_type in {"ce_cost", "focal_loss_cost"}

    @torch.no_grad()
    def forward(self, outputs, targets):
        from scipy.optimize import linear_sum_assignment

        bs, num_queries = outputs["pred_logits"].shape[:2]
        if self.cost_class_type == "ce_cost":
            out_prob = outputs["pred_logits"].flatten(0, 1).softmax(-1)
        else:
            out_prob = outputs["pred_logits"].flatten(0, 1).sigmoid()
        out_bbox = outputs["pred_boxes"].flatten(0, 1)
        tgt_ids = torch.cat([v["labels"] for v in targets])
        tgt_bbox = torch.cat([v["boxes"] for v in targets])
        if self.cost_class_type == "ce_cost":
            cost_class = -out_prob[:, tgt_ids]
        else:
            neg = (1 - self.alpha) * (out_prob ** self.gamma) * (-(1 - out_prob + 1e-8).log())
            pos = self.alpha * ((1 - out_prob) ** self.gamma) * (-(out_prob + 1e-8).log())
            cost_class = pos[:, tgt_ids] - neg[:, tgt_ids]
        cost_bbox = torch.cdist(out_bbox, tgt_bbox, p=1)
        cost_giou = -generalized_box_iou(box_cxcywh_to_xyxy(out_bbox), box_cxcywh_to_xyxy(tgt_bbox))
        C = self.cost_bbox * cost_bbox + self.cost_class * cost_class + self.cost_giou * cost_giou
        C = C.view(bs, num_queries, -1).cpu()
        sizes = [len(v["boxes"]) for v in targets]
        indices = [linear_sum_assignment(c[i]) for i, c in enumerate(C.split(sizes, -1))]
        return [(torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64))
                for i, j in indices]


def get_world_size():
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return 1
    return dist.get_world_size()


def is_dist_avail_and_initialized():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


# --------------------------------------------------------------------------------------
# detectron2 structures
# --------------------------------------------------------------------------------------
class Boxes:
    def __init__(self, tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        else:
            tensor = tensor.to(torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def scale(self, scale_x, scale_y):
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    def clip(self, box_size):
        assert torch.isfinite(self.tensor).all(), "Box tensor contains infinite or NaN!"
        h, w = box_size
        x1 = self.tensor[:, 0].clamp(min=0, max=w)
        y1 = self.tensor[:, 1].clamp(min=0, max=h)
        x2 = self.tensor[:, 2].clamp(min=0, max=w)
        y2 = self.tensor[:, 3].clamp(min=0, max=h)
        self.tensor = torch.stack((x1, y1, x2, y2), dim=-1)

    def nonempty(self, threshold=0.0):
        box = self.tensor
        widths = box[:, 2] - box[:, 0]
        heights = box[:, 3] - box[:, 1]
        return (widths > threshold) & (heights > threshold)

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2
        return Boxes(b)

    def __len__(self):
        return self.tensor.shape[0]


class Instances:
    def __init__(self, image_size, **kwargs):
        self._image_size = image_size
        self._fields = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError(name)
        return self._fields[name]

    def set(self, name, value):
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def get_fields(self):
        return self._fields

    def __getitem__(self, item):
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret


class ImageList:  # imported only
    pass


def detector_postprocess(results, output_height, output_width, mask_threshold=0.5):
    scale_x = output_width / results.image_size[1]
    scale_y = output_height / results.image_size[0]
    results = Instances((output_height, output_width), **results.get_fields())
    output_boxes = results.pred_boxes
    output_boxes.scale(scale_x, scale_y)
    output_boxes.clip(results.image_size)
    results = results[output_boxes.nonempty()]
    return results


# --------------------------------------------------------------------------------------
# mmcv bits
# --------------------------------------------------------------------------------------
class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            self._module_dict[name or cls.__name__] = cls
            return cls

        if module is not None:
            return _register(module)
        return _register

    def build(self, cfg, default_args=None):
        args = dict(cfg)
        if default_args is not None:
            for k, v in default_args.items():
                args.setdefault(k, v)
        obj_type = args.pop("type")
        cls = self.get(obj_type) if isinstance(obj_type, str) else obj_type
        if cls is None:
            raise KeyError(f"{obj_type} is not in the {self._name} registry")
        return cls(**args)


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()


def auto_fp16(apply_to=None, out_fp32=False):
    def deco(fn):
        return fn

    return deco


force_fp32 = auto_fp16


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    """timm.models.layers.trunc_normal_ (== torch.nn.init.trunc_normal_ semantics)."""
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


# ---------------------------------------------------------------------------------------------------------------
# metric leaves used by the reference's simvg/apis/test.py (mmdet / torchvision are not in this image)
# ---------------------------------------------------------------------------------------------------------------
def tv_box_area(boxes):
    """torchvision.ops.boxes.box_area: (x2 - x1) * (y2 - y1) over [N, 4] xyxy boxes."""
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def mmdet_bbox_overlaps(bboxes1, bboxes2, mode="iou", is_aligned=False, eps=1e-6):
    """mmdet 2.x `mmdet/core/bbox/iou_calculators/iou2d_calculator.py::bbox_overlaps`, the branches the reference
    reaches (mode='iou'; aligned: IoU of row i with row i; union clamped from below by eps)."""
    assert mode == "iou"
    area1 = (bboxes1[..., 2] - bboxes1[..., 0]) * (bboxes1[..., 3] - bboxes1[..., 1])
    area2 = (bboxes2[..., 2] - bboxes2[..., 0]) * (bboxes2[..., 3] - bboxes2[..., 1])
    if is_aligned:
        lt = torch.max(bboxes1[..., :2], bboxes2[..., :2])
        rb = torch.min(bboxes1[..., 2:], bboxes2[..., 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = area1 + area2 - overlap
    else:
        lt = torch.max(bboxes1[..., :, None, :2], bboxes2[..., None, :, :2])
        rb = torch.min(bboxes1[..., :, None, 2:], bboxes2[..., None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = area1[..., None] + area2[..., None, :] - overlap
    union = torch.max(union, union.new_tensor([eps]))
    return overlap / union


# ---------------------------------------------------------------------------------------------------------------------
# transformers 4.x `XLMRobertaTokenizer` (sentencepiece-backed "slow" tokenizer; transformers 5.x in this image only has
# the `tokenizers`-backed class with another constructor).  Restated from its published algorithm
# (tokenization_xlm_roberta.py): fairseq alignment of the first four ids, spm ids shifted by one, <mask> last.
# Used only to execute the reference's `LoadImageAnnotationsFromFile` (loading.py:74-77,157-182).  Parity unpinned at
# this boundary.
# ---------------------------------------------------------------------------------------------------------------------
class XLMRobertaTokenizer:
    def __init__(self, vocab_file, bos_token="<s>", eos_token="</s>", sep_token="</s>", cls_token="<s>", unk_token="<unk>",
                 pad_token="<pad>", mask_token="<mask>", **kwargs):
        import sentencepiece as spm
        self.sp_model = spm.SentencePieceProcessor()
        self.sp_model.Load(str(vocab_file))
        self.fairseq_tokens_to_ids = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
        self.fairseq_offset = 1
        self.fairseq_tokens_to_ids["<mask>"] = len(self.sp_model) + self.fairseq_offset
        self.fairseq_ids_to_tokens = {v: k for k, v in self.fairseq_tokens_to_ids.items()}
        self.bos_token, self.eos_token, self.pad_token, self.unk_token = bos_token, eos_token, pad_token, unk_token
        self.bos_token_id = self.fairseq_tokens_to_ids[bos_token]
        self.eos_token_id = self.fairseq_tokens_to_ids[eos_token]
        self.pad_token_id = self.fairseq_tokens_to_ids[pad_token]
        self.unk_token_id = self.fairseq_tokens_to_ids[unk_token]

    @property
    def vocab_size(self):
        return len(self.sp_model) + self.fairseq_offset + 1

    def tokenize(self, text):
        return self.sp_model.encode(text, out_type=str)

    def _convert_token_to_id(self, token):
        if token in self.fairseq_tokens_to_ids:
            return self.fairseq_tokens_to_ids[token]
        spm_id = self.sp_model.PieceToId(token)
        return spm_id + self.fairseq_offset if spm_id else self.unk_token_id

    def convert_tokens_to_ids(self, tokens):
        if isinstance(tokens, str):
            return self._convert_token_to_id(tokens)
        return [self._convert_token_to_id(t) for t in tokens]
