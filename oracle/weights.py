"""Seed-driven weights / inputs shared by the oracle, the golden fixtures and the tests.

TEST INFRASTRUCTURE (see oracle/simvg_cpu.py header).  230 M-parameter weights are never
committed: a fixture stores (cfg, seed) and the expected outputs; both sides regenerate
the identical state_dict from the seed with torch's CPU generator (same torch build in the
dev container and on the GPU box).

Key schema == the reference's state_dict.
"""
import math

import torch


def state_dict_spec(cfg):
    """-> ordered list of (key, shape, kind) for MIXDETRMB with the given geometry."""
    D, F_, L, E = cfg.embed_dim, cfg.ffn_dim, cfg.layers, cfg.head_dim
    P, nq = cfg.patch_size, cfg.num_queries
    npatch = (cfg.img_size // P) ** 2
    s = []
    e = "vis_enc.beit3."
    s += [(e + "text_embed.weight", (cfg.vocab_size, D), "emb"),
          (e + "vision_embed.mask_token", (1, 1, D), "tok"),
          (e + "vision_embed.cls_token", (1, 1, D), "tok"),
          (e + "vision_embed.proj.weight", (D, 3, P, P), "w"),
          (e + "vision_embed.proj.bias", (D,), "b"),
          (e + "encoder.embed_positions.A.weight", (npatch + 3, D), "pos"),
          (e + "encoder.embed_positions.B.weight", (1024, D), "pos")]
    for i in range(L):
        l = f"{e}encoder.layers.{i}."
        for proj in ["k_proj", "v_proj", "q_proj", "out_proj"]:
            for m in "AB":
                s += [(f"{l}self_attn.{proj}.{m}.weight", (D, D), "w"), (f"{l}self_attn.{proj}.{m}.bias", (D,), "b")]
        for ln in ["self_attn.inner_attn_ln", "self_attn_layer_norm"]:
            for m in "AB":
                s += [(f"{l}{ln}.{m}.weight", (D,), "g"), (f"{l}{ln}.{m}.bias", (D,), "b")]
        for m in "AB":
            s += [(f"{l}ffn.{m}.fc1.weight", (F_, D), "w"), (f"{l}ffn.{m}.fc1.bias", (F_,), "b"),
                  (f"{l}ffn.{m}.fc2.weight", (D, F_), "w"), (f"{l}ffn.{m}.fc2.bias", (D,), "b"),
                  (f"{l}ffn.{m}.ffn_layernorm.weight", (F_,), "g"), (f"{l}ffn.{m}.ffn_layernorm.bias", (F_,), "b")]
        for m in "AB":
            s += [(f"{l}final_layer_norm.{m}.weight", (D,), "g"), (f"{l}final_layer_norm.{m}.bias", (D,), "b")]
    for m in "AB":
        s += [(f"{e}encoder.layer_norm.{m}.weight", (D,), "g"), (f"{e}encoder.layer_norm.{m}.bias", (D,), "b")]
    h = "head."

    def dec(prefix, n, ffn):
        r = []
        for i in range(n):
            l = f"{prefix}layers.{i}."
            for a in (0, 1):
                r += [(f"{l}attentions.{a}.attn.in_proj_weight", (3 * E, E), "w"),
                      (f"{l}attentions.{a}.attn.in_proj_bias", (3 * E,), "b"),
                      (f"{l}attentions.{a}.attn.out_proj.weight", (E, E), "w"),
                      (f"{l}attentions.{a}.attn.out_proj.bias", (E,), "b")]
            r += [(f"{l}ffns.0.layers.0.0.weight", (ffn, E), "w"), (f"{l}ffns.0.layers.0.0.bias", (ffn,), "b"),
                  (f"{l}ffns.0.layers.1.weight", (E, ffn), "w"), (f"{l}ffns.0.layers.1.bias", (E,), "b")]
            for k in range(3):
                r += [(f"{l}norms.{k}.weight", (E,), "g"), (f"{l}norms.{k}.bias", (E,), "b")]
        r += [(f"{prefix}post_norm_layer.weight", (E,), "g"), (f"{prefix}post_norm_layer.bias", (E,), "b")]
        return r

    s += dec(h + "transformer.decoder.", cfg.dec_layers, cfg.dec_ffn)
    s += [(h + "input_proj.weight", (E, D, 1, 1), "w"), (h + "input_proj.bias", (E,), "b"),
          (h + "input_text_proj.weight", (E, D), "w"), (h + "input_text_proj.bias", (E,), "b"),
          (h + "input_cls_proj.weight", (E, D), "w"), (h + "input_cls_proj.bias", (E,), "b"),
          (h + "query_embed.weight", (nq, E), "pos"),
          (h + "mlp.layers.0.weight", (E, E), "w"), (h + "mlp.layers.0.bias", (E,), "b")]
    for br in ["decoder", "token"]:
        s += [(f"{h}class_embed_{br}.weight", (cfg.num_classes + 1, E), "w"),
              (f"{h}class_embed_{br}.bias", (cfg.num_classes + 1,), "b")]
        for k, (o, i_) in enumerate([(E, E), (E, E), (4, E)]):
            s += [(f"{h}bbox_embed_{br}.layers.{k}.weight", (o, i_), "w"), (f"{h}bbox_embed_{br}.layers.{k}.bias", (o,), "b")]
    s += dec(h + "text_guided_query_generation_transformer.", cfg.tgqg_layers, cfg.tgqg_ffn)
    s += [(h + "criterion.empty_weight", (2,), "eos"), (h + "criterion_harddistill.empty_weight", (2,), "eos")]
    return s


def golden_state_dict(cfg, seed):
    """'Interesting' weights (non-trivial LN affine, non-zero biases, O(1) attention logits) so
    that wrong-expert / wrong-mask / transposed-operand bugs are visible in the outputs."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape, kind in state_dict_spec(cfg):
        if kind == "w":
            fan_in = math.prod(shape[1:])
            t = torch.randn(shape, generator=g) * fan_in ** -0.5
        elif kind == "b":
            t = torch.randn(shape, generator=g) * 0.05
        elif kind == "g":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif kind == "emb":
            t = torch.randn(shape, generator=g) * 0.5
        elif kind == "pos":
            t = torch.randn(shape, generator=g) * 0.3
        elif kind == "tok":
            t = torch.randn(shape, generator=g) * 0.2
        elif kind == "eos":
            t = torch.tensor([1.0, cfg.eos_coef])
        sd[key] = t
    return sd


def reference_init_state_dict(cfg, seed):
    """The distributions of the reference's OWN initialisation: BEiT3Wrapper._init_weights
    (modeling_utils.py:102-109: trunc-normal(0.02, clipped at +-0.02) Linear weights, zero biases, unit LayerNorm),
    torchscale embedding inits (text N(0, D^-1/2), positions N(0,1), zero cls/mask tokens), nn.Conv2d / nn.Linear
    defaults, xavier for the DETR decoder (transformer.py:200-203), nn.Embedding N(0,1) for query_embed.
    Seeded here (not RNG-identical to the reference constructor) so that fixtures are regenerable."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def uni(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    for key, shape, kind in state_dict_spec(cfg):
        enc = key.startswith("vis_enc.")
        if kind == "eos":
            t = torch.tensor([1.0, cfg.eos_coef])
        elif kind == "g":
            t = torch.ones(shape)
        elif kind == "tok":
            t = torch.zeros(shape)
        elif kind == "emb":
            t = torch.randn(shape, generator=g) * cfg.embed_dim ** -0.5
        elif kind == "pos":
            t = torch.randn(shape, generator=g)
        elif kind == "b":
            if enc and "vision_embed.proj" not in key:
                t = torch.zeros(shape)
            elif "in_proj_bias" in key or "out_proj.bias" in key:
                t = torch.zeros(shape)
            else:
                fan_in = {"vis_enc.beit3.vision_embed.proj.bias": 3 * cfg.patch_size ** 2}.get(key)
                if fan_in is None:
                    wshape = dict((k, s) for k, s, _ in state_dict_spec(cfg))[key[:-4] + "weight"]
                    fan_in = math.prod(wshape[1:])
                t = uni(shape, fan_in ** -0.5)
        elif kind == "w":
            fan_in = math.prod(shape[1:])
            if enc and "vision_embed.proj" not in key:
                t = (torch.randn(shape, generator=g) * 0.02).clamp_(-0.02, 0.02)
            elif key.startswith("head.transformer.decoder.") or "in_proj_weight" in key:
                fan_out = shape[0]
                t = uni(shape, math.sqrt(6.0 / (fan_in + fan_out)))      # xavier_uniform
            else:
                t = uni(shape, fan_in ** -0.5)                           # kaiming_uniform(a=sqrt(5))
        sd[key] = t
    return sd


def synthetic_batch(cfg, B, seed, grec=False):
    """RefCOCO-shape synthetic inputs: fp32 image N(0,1), XLM-R-style ids
    [0, t1..tm, 2, 1...] with m~U{2..10}, int64 pad mask (1 = pad, quirk Q1), xyxy pixel gt boxes."""
    g = torch.Generator().manual_seed(seed)
    S, T = cfg.img_size, cfg.max_token
    img = torch.randn(B, 3, S, S, generator=g)
    ids = torch.ones(B, T, dtype=torch.int64)
    pad = torch.ones(B, T, dtype=torch.int64)
    for b in range(B):
        m = int(torch.randint(2, 11, (1,), generator=g))
        ids[b, 0] = 0
        ids[b, 1:1 + m] = torch.randint(4, cfg.vocab_size, (m,), generator=g)
        ids[b, 1 + m] = 2
        pad[b, :m + 2] = 0
    metas, gts = [], []
    for b in range(B):
        meta = dict(img_shape=(S, S, 3), pad_shape=(S, S, 3), ori_shape=(S, S, 3),
                    scale_factor=[1.0, 1.0, 1.0, 1.0], filename=f"synthetic_{b}.jpg", expression="synthetic")
        k = 1
        if grec:
            k = int(torch.randint(1, 4, (1,), generator=g))
        xy = torch.rand(k, 2, generator=g) * (S * 0.625)
        wh = S * 0.05 + torch.rand(k, 2, generator=g) * (S * 0.325)
        box = torch.cat([xy, torch.minimum(xy + wh, torch.tensor(float(S)))], dim=1)
        if grec:
            if b % 3 == 1:  # a no-target sample (loading.py:224-239)
                box = torch.zeros(1, 4)
                meta["target"] = [{"category_id": -1}]
            else:
                meta["target"] = [{"category_id": 1} for _ in range(k)]
            gts.append(box)
        else:
            gts.append(box[0])
        metas.append(meta)
    return dict(img=img, ref_expr_inds=ids, text_attention_mask=pad, img_metas=metas, gt_bbox=gts)
