"""TextGuidedQuerySelectKDDETRHead on MI355X -- the decoupled head of SimVG behind the reference API.

Host-side mirror of `simvg/models/heads/tgqs_kd_detr_head/tgqs_kd_detr_head.py:22-604` (registered in HEADS,
same constructor kwargs, `forward_train` / `forward_test` / `inference`), of the DETR decoder in
`transformer.py:93-235` and of `core/criterion/criterion.py:62-271`, with all arithmetic in hand-written
gfx950 kernels: 16-bit MFMA GEMMs for the B*(1+HW) memory rows, exact-fp32 MFMA GEMMs / LayerNorm / attention
for the [B*nq, 256] query rows, and an on-device Hungarian matcher + criterion (no host round trips).
Only the branches the reference configs execute are built: `branch_loss_weight` with "decoder" and / or
"balanced_distill" (32 configs use both; the 13 two-stage stage-1, 2 pre-training and 6 fine-tuning configs use
{"decoder": 1.0} alone, `tgqs_kd_detr_head.py:483-509`: independent `if` blocks), `hard_weighted`,
`score_iou_weighted`, TGQG on; SURVEY.md 8(a) "dead branches".  state_dict keys match Appendix B.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import builder
from .... import hip_ops as ops
from ..functions import (Criterion, DecoderLayerFn, DecoderLayerUnfusedFn, LayerCfg, PredHeadFn, QueryMix, TextFilt, LayerNormF32, LinearLP, LinearF32, SharedMemoryGrad,
                         SplitEncoderOutput)


def _xavier(*shape):
    t = torch.empty(*shape)
    if not builder.skip_init.active:
        nn.init.xavier_uniform_(t)
    return t


def _linear_default(out_f, in_f):   # nn.Linear default init
    w = torch.empty(out_f, in_f)
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    bound = 1 / math.sqrt(in_f)
    return w, torch.empty(out_f).uniform_(-bound, bound)


class _Bag(nn.Module):
    pass


def _set(root, key, tensor):
    parts = key.split(".")
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, _Bag())
        m = m._modules[p]
    m.register_parameter(parts[-1], nn.Parameter(tensor))


def sine_pos_2d(mask, num_pos_feats=128, temperature=10000, scale=2 * math.pi, eps=1e-6):
    """detrex PositionEmbeddingSine(normalize=True): mask [B,H,W] bool -> [B, H*W, 2*num_pos_feats]."""
    not_mask = ~mask
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32, device=mask.device)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    px = x_embed[..., None] / dim_t
    py = y_embed[..., None] / dim_t
    B, H, W = mask.shape
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).view(B, H, W, -1)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).view(B, H, W, -1)
    return torch.cat((py, px), dim=3).flatten(1, 2)


def sine_pos_1d(pos_len, dim):
    """PositionEmbeddingSine1D incl. quirk Q2 (frequencies cast to long -> [1,0,0,...]); heads/utils.py:72-100."""
    i = torch.arange(dim // 2, dtype=torch.float)
    i = (1 / torch.pow(10000, i / (dim / 2))).to(torch.long)
    out = torch.arange(pos_len).to(torch.long)[:, None] @ i[None, :]
    emb = torch.zeros(pos_len, dim)
    emb[:, 0::2] = torch.sin(out)
    emb[:, 1::2] = torch.cos(out)
    return emb


@builder.HEADS.register_module()
class TextGuidedQuerySelectKDDETRHead(nn.Module):
    def __init__(self, num_queries=100, in_channels=768, text_max_token=20, embed_dim=256, num_classes=1,
                 aux_loss=True, num_encoder_layers=6, num_decoder_layers=6, num_tgqg_layers=1, only_decoder=False,
                 text_embed_aug=False, branch_loss_weight={}, as_target_query_thr=0.0, distill_type="",
                 decoder_freeze=False, prepare_target_mode="score_weighted", share_predicthead=False,
                 num_token_mlp_layers=3, mlp_aux_loss=False, tgqs_mid_dim=512, aux_distill_mode="klloss",
                 text_guided_query_generation=False):
        super().__init__()
        assert prepare_target_mode in ["score_weighted", "score_iou_weighted"]
        assert distill_type in ["hard", "hard_weighted", "soft"]
        assert all(x in ["decoder", "token", "distill", "merge", "aux_distill", "balanced_distill"]
                   for x in branch_loss_weight.keys())
        unsupported = []
        if not only_decoder: unsupported.append("only_decoder=False")
        if not text_guided_query_generation: unsupported.append("text_guided_query_generation=False")
        if prepare_target_mode != "score_iou_weighted": unsupported.append("prepare_target_mode=" + prepare_target_mode)
        if not branch_loss_weight or not set(branch_loss_weight) <= {"decoder", "balanced_distill"}:
            unsupported.append(f"branch_loss_weight={set(branch_loss_weight)}")
        if num_token_mlp_layers != 1 or mlp_aux_loss or share_predicthead or decoder_freeze or not aux_loss or num_classes != 1:
            unsupported.append("token-MLP / predict-head / aux options")
        if embed_dim != 256:
            unsupported.append("embed_dim != 256")
        if unsupported:
            raise NotImplementedError("simvg_amd builds the hot path of the reference configs only; not built: "
                                      + ", ".join(unsupported))
        if num_queries > 16:
            raise NotImplementedError("on-device matcher / query kernels are built for num_queries <= 16")
        self.num_queries, self.in_channels, self.embed_dim = num_queries, in_channels, embed_dim
        self.num_classes, self.aux_loss = num_classes, aux_loss
        self.num_decoder_layers, self.num_tgqg_layers = num_decoder_layers, num_tgqg_layers
        self.branch_loss_weight = branch_loss_weight
        self.dec_ffn, self.tgqg_ffn, self.heads = 2048, tgqs_mid_dim, 8
        self.attn_dropout = self.ffn_dropout = 0.1
        self.eos_coef = 0.1
        self.cost = (1.0, 5.0, 2.0)            # HungarianMatcher(cost_class, cost_bbox, cost_giou) :132-137
        self.loss_w = (1.0, 5.0, 2.0)          # weight_dict loss_class / loss_bbox / loss_giou :141-145
        self.max_targets = 16
        self._build_parameters()
        self.register_buffer("criterion_empty_weight", torch.tensor([1.0, self.eos_coef]), persistent=False)
        self._prep = None
        self._prep_version = None
        self._mask_state, self._mask_plan = None, {}
        self._const = {}

    # ------------------------------------------------------------------ parameters (reference schema)
    def _build_parameters(self):
        E, C, nq = self.embed_dim, self.in_channels, self.num_queries
        r = self

        def dec(prefix, n, ffn):
            for i in range(n):
                l = f"{prefix}layers.{i}."
                for a in (0, 1):   # DetrTransformer.init_weights: xavier on every dim>1 param (transformer.py:200-203)
                    _set(r, f"{l}attentions.{a}.attn.in_proj_weight", _xavier(3 * E, E))
                    _set(r, f"{l}attentions.{a}.attn.in_proj_bias", torch.zeros(3 * E))
                    _set(r, f"{l}attentions.{a}.attn.out_proj.weight", _xavier(E, E))
                    _set(r, f"{l}attentions.{a}.attn.out_proj.bias", torch.zeros(E))
                w1, b1 = _linear_default(ffn, E)
                w2, b2 = _linear_default(E, ffn)
                _set(r, f"{l}ffns.0.layers.0.0.weight", _xavier(ffn, E)); _set(r, f"{l}ffns.0.layers.0.0.bias", b1)
                _set(r, f"{l}ffns.0.layers.1.weight", _xavier(E, ffn)); _set(r, f"{l}ffns.0.layers.1.bias", b2)
                for k in range(3):
                    _set(r, f"{l}norms.{k}.weight", torch.ones(E)); _set(r, f"{l}norms.{k}.bias", torch.zeros(E))
            _set(r, f"{prefix}post_norm_layer.weight", torch.ones(E)); _set(r, f"{prefix}post_norm_layer.bias", torch.zeros(E))

        dec("transformer.decoder.", self.num_decoder_layers, self.dec_ffn)
        w, b = _linear_default(E, C)
        _set(r, "input_proj.weight", w.view(E, C, 1, 1)); _set(r, "input_proj.bias", b)
        for name in ["input_text_proj", "input_cls_proj"]:
            w, b = _linear_default(E, C)
            _set(r, name + ".weight", w); _set(r, name + ".bias", b)
        _set(r, "query_embed.weight", torch.randn(nq, E))
        w, b = _linear_default(E, E)
        _set(r, "mlp.layers.0.weight", w); _set(r, "mlp.layers.0.bias", b)
        for br in ["decoder", "token"]:
            w, b = _linear_default(self.num_classes + 1, E)
            _set(r, f"class_embed_{br}.weight", w); _set(r, f"class_embed_{br}.bias", b)
            for k, (o, i_) in enumerate([(E, E), (E, E), (4, E)]):
                w, b = _linear_default(o, i_)
                _set(r, f"bbox_embed_{br}.layers.{k}.weight", w); _set(r, f"bbox_embed_{br}.layers.{k}.bias", b)
        # the TGQG decoder is NOT re-initialised by DetrTransformer.init_weights (it is a sibling module):
        # nn.MultiheadAttention default = xavier in_proj, zero biases, default Linear out_proj / FFN
        for i in range(self.num_tgqg_layers):
            l = f"text_guided_query_generation_transformer.layers.{i}."
            for a in (0, 1):
                _set(r, f"{l}attentions.{a}.attn.in_proj_weight", _xavier(3 * E, E))
                _set(r, f"{l}attentions.{a}.attn.in_proj_bias", torch.zeros(3 * E))
                w, _ = _linear_default(E, E)
                _set(r, f"{l}attentions.{a}.attn.out_proj.weight", w); _set(r, f"{l}attentions.{a}.attn.out_proj.bias", torch.zeros(E))
            w1, b1 = _linear_default(self.tgqg_ffn, E)
            w2, b2 = _linear_default(E, self.tgqg_ffn)
            _set(r, f"{l}ffns.0.layers.0.0.weight", w1); _set(r, f"{l}ffns.0.layers.0.0.bias", b1)
            _set(r, f"{l}ffns.0.layers.1.weight", w2); _set(r, f"{l}ffns.0.layers.1.bias", b2)
            for k in range(3):
                _set(r, f"{l}norms.{k}.weight", torch.ones(E)); _set(r, f"{l}norms.{k}.bias", torch.zeros(E))
        _set(r, "text_guided_query_generation_transformer.post_norm_layer.weight", torch.ones(E))
        _set(r, "text_guided_query_generation_transformer.post_norm_layer.bias", torch.zeros(E))

    # the reference state_dict also carries the two criterion buffers
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        ew = self.criterion_empty_weight if keep_vars else self.criterion_empty_weight.detach()
        sd[prefix + "criterion.empty_weight"] = ew
        sd[prefix + "criterion_harddistill.empty_weight"] = ew.clone()
        return sd

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for k in ("criterion.empty_weight", "criterion_harddistill.empty_weight"):
            state_dict.pop(prefix + k, None)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    # ------------------------------------------------------------------ 16-bit copies of the memory-row weights
    def _P(self, key):
        m = self
        for p in key.split("."):
            m = getattr(m, p)
        return m

    def _refresh_weights(self, device):
        E, C = self.embed_dim, self.in_channels
        # (round 5: the decoder's cross-attention contracts the memory rows directly -- csrc/decoder.hip --, so input_proj is the
        # only Linear of the head that runs on the 16-bit MFMA GEMM and needs 16-bit weight copies)
        srcs = [self._P("input_proj.weight")]
        version = tuple(p._version for p in srcs) + (srcs[0].data_ptr(),)
        if self._prep is not None and version == self._prep_version and not self.training:
            return
        if self._prep is None or self._prep_dev != device or self._prep_ptrs != [p.data_ptr() for p in srcs]:
            def bf(*s):
                return torch.empty(*s, device=device, dtype=ops.LP())
            self.wb = {"ip": bf(E, C), "ipT": bf(C, E)}
            entries = [(srcs[0].data.view(E, C), self.wb["ip"], self.wb["ipT"])]
            self._prep = ops.WeightPrep(entries, device)
            self._prep_dev, self._prep_ptrs = device, [p.data_ptr() for p in srcs]
        self._prep.run()
        # training: the optimizer may rewrite the parameters without moving their version counters (FlatAdam updates them
        # through one flat tensor) -> the first eval forward after training always refreshes
        self._prep_version = None if self.training else version

    def mark_weights_dirty(self):
        self._prep_version = None

    # ------------------------------------------------------------------ building blocks
    def _lin(self, x, key, relu=False, rows=None):
        W, b = self._P(key + ".weight"), self._P(key + ".bias")
        if rows is not None:
            W, b = W[rows[0]:rows[1]], b[rows[0]:rows[1]]
        shp = x.shape
        y = LinearF32.apply(x.reshape(-1, shp[-1]), W.view(W.shape[0], -1), b, relu)
        return y.view(*shp[:-1], -1)

    def _pred_head(self, x, branch, lead=None):
        """class_embed_<branch> and bbox_embed_<branch> (+ sigmoid) on the rows x [M, E] as one autograd node (PredHeadFn); lead: key
        of a Linear applied first (the token branch's mlp)"""
        P = self._P
        Wm, bm = (P(lead + ".weight"), P(lead + ".bias")) if lead else (None, None)
        w = [P(f"bbox_embed_{branch}.layers.{k}.{n}") for k in range(3) for n in ("weight", "bias")]
        return PredHeadFn.apply(x, Wm, bm, P(f"class_embed_{branch}.weight"), P(f"class_embed_{branch}.bias"), *w)

    def _ln(self, x, key):
        shp = x.shape
        return LayerNormF32.apply(x.reshape(-1, shp[-1]), self._P(key + ".weight"), self._P(key + ".bias"), 1e-5).view(shp)

    def _drop_mult(self, shape, device, p=None):
        """dropout multipliers (0 or 1/(1-p)) of one site.  All sites of a forward are slices of ONE generated buffer per
        rate: the first forward of a geometry records how much each rate needs (`_mask_plan`), later ones draw the whole
        step's multipliers in a single launch (`_begin_masks`) -- 20 launches per step otherwise."""
        p = self.attn_dropout if p is None else p
        if not self.training or p == 0:
            return None
        n = 1
        for d in shape:
            n *= int(d)
        n_al = (n + 63) // 64 * 64
        st = self._mask_state
        if st is not None:
            st["need"][p] = st["need"].get(p, 0) + n_al
            buf = st["buf"].get(p)
            if buf is not None and st["off"].get(p, 0) + n_al <= buf.numel():
                o = st["off"].get(p, 0)
                st["off"][p] = o + n_al
                return buf[o:o + n].view(shape)
        return ops.dropout_mult(n, device, keep=1.0 - p).view(shape)

    def _begin_masks(self, key, device):
        if not self.training:
            self._mask_state = None
            return
        plan = self._mask_plan.get(key, {})
        bufs = {}
        for p, total in plan.items():
            bufs[p] = ops.dropout_mult(total, device, keep=1.0 - p)       # csrc/rng.hip: one launch for all sites of the step
        self._mask_state = dict(key=key, need={}, off={}, buf=bufs)

    def _end_masks(self):
        st, self._mask_state = self._mask_state, None
        if st is not None and len(self._mask_plan) < 64:
            self._mask_plan[st["key"]] = dict(st["need"])

    _LAYER_KEYS = ("attentions.0.attn.in_proj_weight", "attentions.0.attn.in_proj_bias", "attentions.0.attn.out_proj.weight",
                   "attentions.0.attn.out_proj.bias", "norms.0.weight", "norms.0.bias",
                   "attentions.1.attn.in_proj_weight", "attentions.1.attn.in_proj_bias", "attentions.1.attn.out_proj.weight",
                   "attentions.1.attn.out_proj.bias", "norms.1.weight", "norms.1.bias",
                   "ffns.0.layers.0.0.weight", "ffns.0.layers.0.0.bias", "ffns.0.layers.1.weight", "ffns.0.layers.1.bias",
                   "norms.2.weight", "norms.2.bias")

    def _decoder_layer(self, L, tgt, qpos, cfg, src, post=None):
        """BaseTransformerLayer, post-norm order (self_attn, norm, cross_attn, norm, ffn, norm) as one autograd node;
        tgt / qpos [B*nq, E]; src: the cross-attention's source rows (LayerCfg); `post` = key prefix of a LayerNorm applied to
        the layer output (the decoder's shared post_norm_layer) -> returns (layer output, post-normed output | None)."""
        params = [self._P(L + k) for k in self._LAYER_KEYS]
        gP, bP = (self._P(post + ".weight"), self._P(post + ".bias")) if post else (None, None)
        # the fused layer (csrc/decoder.hip: a workgroup owns a sample's query rows) is the faster one for ONE query per sample (the
        # RefCOCO configs, the benchmark): measured at 64 samples x 10 queries it loses to the per-stage kernels (37.6 vs 32.2 ms per
        # step: 640 rows on 64 workgroups, the FFN's hidden slices walk them serially) -- profiles/r05_sweeps.md.
        # SIMVG_DEC_UNFUSED=1 / SIMVG_DEC_FUSED=1 force one form (A/B measurements, tests of the multi-query kernels)
        env_unfused, env_fused = os.environ.get("SIMVG_DEC_UNFUSED") == "1", os.environ.get("SIMVG_DEC_FUSED") == "1"
        if cfg.Lk > ops.dec_attn_max_keys() or env_unfused or (cfg.nq > 1 and not env_fused):
            if tgt is None:
                tgt = torch.zeros_like(qpos)
            # more keys than the fused kernels hold in LDS (patch 16 at 480 / 640 px; no reference config): the per-stage kernels
            if cfg.kind == "text":
                xk = (src.view(cfg.B, cfg.Lk, -1) + cfg.pos[None]).reshape(src.shape)
                return DecoderLayerUnfusedFn.apply(tgt, qpos, xk, src, None, *params, gP, bP, cfg)
            if src.dtype == ops.LP():
                cfg.wb, cfg.wbT = self._kv_weights(L, src.device)
            return DecoderLayerUnfusedFn.apply(tgt, qpos, None, None, src, *params, gP, bP, cfg)
        return DecoderLayerFn.apply(tgt, qpos, src, *params, gP, bP, cfg)

    def _kv_weights(self, L, device):
        """16-bit copies (plain, transposed) of a layer's cross-attention K|V in-projection rows, for the unfused layer; refreshed
        on every call (that path is not a hot one)"""
        E = self.embed_dim
        W = self._P(L + "attentions.1.attn.in_proj_weight")
        key = ("kv", L)
        ent = self._const.get(key)
        if ent is None or ent[0].device != device or ent[3] != W.data_ptr():
            wb, wbT = torch.empty(2 * E, E, device=device, dtype=ops.LP()), torch.empty(E, 2 * E, device=device, dtype=ops.LP())
            ent = self._const[key] = (wb, wbT, ops.WeightPrep([(W.data[E:], wb, wbT)], device), W.data_ptr())
        ent[2].run()
        return ent[0], ent[1]

    def _layer_cfg(self, B, kind, Lk, **kw):
        return LayerCfg(B, self.heads, self.num_queries, kind, Lk, p_attn=self.attn_dropout, p_ffn=self.ffn_dropout,
                        training=self.training, mask_fn=self._drop_mult, **kw)

    def _constants(self, device, T, hw):
        key = (str(device), T, hw)
        c = self._const.get(key)
        if c is None:
            c = dict(tpos=sine_pos_1d(T, self.embed_dim).to(device))
            self._const[key] = c
        return c

    def _image_pos(self, B, hw, img_metas, device):
        """x_mask_pos_enc (:322-338).  Fast path: every image fills the batch canvas -> constant embedding."""
        try:
            Hin, Win = img_metas[0]["batch_input_shape"]
        except Exception:
            Hin, Win, _ = img_metas[0]["img_shape"]
        full = all(tuple(m["img_shape"][:2]) == (Hin, Win) for m in img_metas)
        if full:
            key = ("pos2d", str(device), hw)
            if key not in self._const:
                self._const[key] = sine_pos_2d(torch.zeros(1, hw, hw, dtype=torch.bool, device=device), self.embed_dim // 2)[0]
            return self._const[key], None
        m = torch.ones((B, Hin, Win), device=device)
        for i in range(B):
            h, w = img_metas[i]["img_shape"][:2]
            m[i, :h, :w] = 0
        mask = F.interpolate(m.unsqueeze(1), size=(hw, hw)).to(torch.bool).squeeze(1)
        return sine_pos_2d(mask, self.embed_dim // 2), mask.flatten(1).to(torch.uint8).contiguous()

    # ------------------------------------------------------------------ forward_general (:375-454)
    def forward_fused(self, enc_out, B, Nv, T, img_metas, text_mask):
        """enc_out: encoder output [B*Nv + B*T, D] fp32, modality-major (BEIT3.encode); with a 16-bit copy of the same
        rows as `enc_out.lp` the memory projections run on the 16-bit MFMA GEMMs, without it everything is exact fp32."""
        device = enc_out.device
        E, nq, H = self.embed_dim, self.num_queries, self.heads
        HW = Nv - 1
        self._begin_masks((str(device), B, Nv, T), device)
        hw = int(round(HW ** 0.5))
        enc_lp = getattr(enc_out, "lp", None)
        exact = enc_lp is None                      # precision="fp32" mode: fp32 memory rows as well
        C = self.in_channels
        if exact:
            vis, text32 = enc_out[:B * Nv], enc_out[B * Nv:]
            cls32 = vis.view(B, Nv, C)[:, 0]
            mem = LinearF32.apply(vis, self._P("input_proj.weight").view(E, C), self._P("input_proj.bias"), False)
        else:
            self._refresh_weights(device)
            vis_holder = {}
            vis, text32, cls32 = SplitEncoderOutput.apply(enc_out, enc_lp, B, Nv, T, vis_holder)
            # H1: input_proj on all vision rows (the CLS row is carried along and never used as a key)
            mem_holder = {}
            mem = LinearLP.apply(vis, self._P("input_proj.weight"), self._P("input_proj.bias"), self.wb["ip"], self.wb["ipT"], True, mem_holder, vis_holder)
        text = self._lin(text32, "input_text_proj")                       # [B*T, E]
        cls = self._lin(cls32, "input_cls_proj")                          # [B, E]
        pos2d, img_kpm = self._image_pos(B, hw, img_metas, device)
        c = self._constants(device, T, hw)
        # ---- TGQG (:385-399)
        text3 = text.view(B, T, E)
        if text_mask.dtype == torch.bool:
            filt = text3.masked_fill(text_mask[:, :, None], float("-inf")).max(1)[0]
            tkpm = text_mask.to(torch.uint8).contiguous()
        else:   # Q1: `~mask` on an int64 mask is a bitwise NOT -> rows T-1 (mask==0 present) and T-2 (mask==1 present)
            filt, tkpm = TextFilt.apply(text, text_mask.to(torch.int64).contiguous(), B, T)
        qe = self._P("query_embed.weight")
        qpos = qe.unsqueeze(0).expand(B, nq, E).reshape(B * nq, E)

        tgt = None                                      # zeros (transformer.py:220: `target = torch.zeros_like(query_embed)`)
        pre = "text_guided_query_generation_transformer."
        # the TGQG layers attend from the queries to the text rows: keys = text + 1-D sine positions, values = text (:391-399)
        cfg_t = self._layer_cfg(B, "text", T, kpm=tkpm, pos=c["tpos"])
        g = None
        for i in range(self.num_tgqg_layers):
            last = i == self.num_tgqg_layers - 1
            tgt, g = self._decoder_layer(f"{pre}layers.{i}.", tgt, qpos, cfg_t, text,
                                         post=pre + "post_norm_layer" if last else None)
        query_embed, tok = QueryMix.apply(g, filt, qe, cls, B, nq)                    # g + filt + qe ; + cls (Q5)
        query_embed = query_embed.view(B, nq, E)
        # ---- token branch (:411-420); with branch_loss_weight == {"decoder": w} the reference skips it, in forward_test
        # as well: the token prediction is {"pred_logits": None, "pred_boxes": None} and `token_features` the pre-MLP sum (:403-409)
        if set(self.branch_loss_weight) == {"decoder"}:
            tok_logits = tok_boxes = None
        else:
            tl, tbx, tok = self._pred_head(tok, "token", lead="mlp.layers.0")
            tok_logits, tok_boxes = tl.view(1, B, nq, -1), tbx.view(1, B, nq, 4)
        # ---- decoder branch (:425-428)
        qpos_d = query_embed.reshape(B * nq, E)
        tgt = None                                      # zeros (transformer.py:220: `target = torch.zeros_like(query_embed)`)
        hs = []
        mem_grad = None
        if not exact and mem.requires_grad and torch.is_grad_enabled():
            mem, mem_grad = SharedMemoryGrad.join(mem, mem_holder)     # one fp32 accumulator for the layers' memory gradients
        for i in range(self.num_decoder_layers):
            # keys = mem + pos, values = mem (key_pos on the keys only); the K / V projections are absorbed into the query / output side
            cfg_m = self._layer_cfg(B, "mem", HW, kpm=img_kpm, pos=pos2d, Nv=Nv, mem_grad=mem_grad)
            tgt, h = self._decoder_layer(f"transformer.decoder.layers.{i}.", tgt, qpos_d, cfg_m, mem,
                                         post="transformer.decoder.post_norm_layer")
            hs.append(h)
        hs = torch.stack(hs).view(self.num_decoder_layers, B, nq, E)
        dl, dbx, _ = self._pred_head(hs.reshape(-1, E), "decoder")
        dec_logits, dec_boxes = dl.view(self.num_decoder_layers, B, nq, -1), dbx.view(self.num_decoder_layers, B, nq, 4)
        self._end_masks()
        return dict(
            token_branch_output={"pred_logits": None if tok_logits is None else tok_logits[-1],
                                 "pred_boxes": None if tok_boxes is None else tok_boxes[-1]},
            decoder_branch_output={"pred_logits": dec_logits[-1], "pred_boxes": dec_boxes[-1]},
            outputs_class_decoder_branch=dec_logits, outputs_coord_decoder_branch=dec_boxes,
            outputs_class_token_branch=tok_logits, outputs_coord_token_branch=tok_boxes,
            token_features=tok.view(1, B, nq, E) if tok_logits is not None else tok.view(B, nq, E), decoder_features=hs)

    # ------------------------------------------------------------------ targets + losses (:207-268, 456-572)
    def _pack_targets(self, gt_bbox, img_metas, device, return_counts=False):
        """GT -> normalised cxcywh target arrays (prepare_soft_targets :215-234; drops category_id == -1 entries).
        Counts / indices come from host metadata (shapes, img_metas); boxes in HBM are read where they are (the table carries
        their addresses), host boxes travel inside the table: one host->device copy and one launch (`simvg_pack_targets`), no
        device-to-host synchronisation."""
        B, TM = len(gt_bbox), self.max_targets
        rows, counts = self._target_rows(gt_bbox, img_metas, device)
        boxes, count = ops.pack_targets(rows, counts, B, TM, device)
        key = ("tlabels", str(device), B, TM)
        labels = self._const.get(key)
        if labels is None:
            labels = self._const[key] = torch.zeros(B, TM, dtype=torch.int32, device=device)
        out = (boxes, labels, count)
        return out + (counts,) if return_counts else out

    def _target_rows(self, gt_bbox, img_metas, device):
        """host side of the packing: -> (table rows for `hip_ops.pack_targets`, number of kept targets per image)"""
        TM = self.max_targets
        rows, counts = [], []
        for b, (tb, meta) in enumerate(zip(gt_bbox, img_metas)):
            tb = tb if torch.is_tensor(tb) else torch.as_tensor(tb, dtype=torch.float32)
            h, w = meta["img_shape"][:2]
            if tb.dim() == 1:
                tb = tb.unsqueeze(0)
                keep = [0]
            else:
                assert int(tb.shape[0]) == len(meta["target"])
                keep = [i for i, t in enumerate(meta["target"]) if t["category_id"] != -1]
            if len(keep) > TM:
                raise ValueError(f"more than {TM} targets in one image")
            if keep:
                on_dev = tb.is_cuda and tb.device == device and tb.dtype == torch.float32 and tb.stride(-1) == 1
                if not on_dev:            # host boxes (the file loader's) travel inside the table; other device layouts are normalised first
                    vals = tb.detach().to(dtype=torch.float32).cpu().tolist() if not tb.is_cuda else None
                    if vals is None:
                        tb, on_dev = tb.to(device=device, dtype=torch.float32).contiguous(), True
                for j, i in enumerate(keep):
                    if on_dev:
                        rows.append((tb, i * tb.stride(0), None, w, h, b * TM + j))      # (tensor, element offset of the row)
                    else:
                        rows.append((None, 0, vals[i], w, h, b * TM + j))
            counts.append(len(keep))
        return rows, counts

    def prepare_targets(self, gt_bbox, img_metas, device):
        """Everything the criterion needs that is known BEFORE the forward: packed targets and the two loss normalisers
        num_boxes = max(sum_ranks(k) / world, 1) (criterion.py:245-249, C3) for the GT target set (k = #GT boxes) and
        for the pseudo-target set of the KD term (k = #matched queries = min(num_queries, #GT) per image).  Keeping this
        outside `loss_from_targets` leaves the latter free of collectives and host copies (CUDA-graph capturable)."""
        tboxes, tlabels, tcount, counts = self._pack_targets(gt_bbox, img_metas, device, return_counts=True)
        return tboxes, tlabels, tcount, self.target_normalisers(counts, device)

    def target_normalisers(self, counts, device):
        """[num_boxes of the GT target set, of the KD pseudo-target set], averaged over the ranks (criterion.py:245-249)"""
        n_gt = float(sum(counts))
        n_kd = float(sum(min(self.num_queries, c) for c in counts))
        key = ("nums", str(device), n_gt, n_kd)
        nums = self._const.get(key)
        if nums is None:
            nums = torch.tensor([n_gt, n_kd], dtype=torch.float32).to(device)
            if len(self._const) < 4096:
                self._const[key] = nums
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            nums = nums.clone()
            torch.distributed.all_reduce(nums)
            nums = nums / torch.distributed.get_world_size()
        return nums

    def loss(self, output, gt_bbox, img_metas):
        tboxes, tlabels, tcount, nums = self.prepare_targets(gt_bbox, img_metas, output["outputs_class_decoder_branch"].device)
        return self.loss_from_targets(output, tboxes, tlabels, tcount, nums)

    @property
    def loss_keys(self):
        """keys of forward_train's loss dict, in the reference's insertion order (:483-509, :571)"""
        keys = []
        if "decoder" in self.branch_loss_weight:
            keys.append("loss_dgt")
        if "balanced_distill" in self.branch_loss_weight:
            keys += ["loss_tgt", "loss_kd", "loss_distill_w"]
        return tuple(keys) + ("loss_total",)

    def loss_from_targets(self, output, tboxes, tlabels, tcount, nums):
        """forward_train's loss composition (:483-509, :571): the "decoder" and "balanced_distill" blocks are independent;
        a block that is not configured contributes neither a dict entry nor a gradient (with {"decoder": 1.0} alone the
        token branch -- input_cls_proj, mlp, class_embed_token, bbox_embed_token -- receives none, as in the reference)."""
        dl, dbx = output["outputs_class_decoder_branch"], output["outputs_coord_decoder_branch"]
        dl_d, dbx_d = dl.detach().contiguous(), dbx.detach().contiguous()
        m_dec = ops.match(dl_d, dbx_d, tboxes, tlabels, tcount, self.cost)
        bw = self.branch_loss_weight
        losses, detail, total = {}, dict(match_dec=m_dec, targets=(tboxes, tcount)), None
        if "decoder" in bw:
            loss_dgt, detail["terms_dec"] = Criterion.apply(dl, dbx, m_dec, tboxes, tlabels, nums[0:1], None, 0,
                                                            float(bw["decoder"]), self.eos_coef, self.loss_w)
            losses["loss_dgt"] = total = loss_dgt
        if "balanced_distill" in bw:
            tl, tbx = output["outputs_class_token_branch"][-1:], output["outputs_coord_token_branch"][-1:]
            pboxes, plabels, pcount, pweight, scal = ops.soft_targets(dl_d[-1], dbx_d[-1], m_dec[-1], tboxes, tcount)
            wd = scal[0:1]
            tl_d, tbx_d = tl.detach().contiguous(), tbx.detach().contiguous()
            m_tg = ops.match(tl_d, tbx_d, tboxes, tlabels, tcount, self.cost)
            loss_tgt, t_tg = Criterion.apply(tl, tbx, m_tg, tboxes, tlabels, nums[0:1], wd, 1,
                                             float(bw["balanced_distill"]["token"]), self.eos_coef, self.loss_w)
            m_kd = ops.match(tl_d, tbx_d, pboxes, plabels, pcount, self.cost)
            loss_kd, t_kd = Criterion.apply(tl, tbx, m_kd, pboxes, plabels, nums[1:2], wd, 2,
                                            float(bw["balanced_distill"]["distill"]), self.eos_coef, self.loss_w)
            losses.update(loss_tgt=loss_tgt, loss_kd=loss_kd, loss_distill_w=wd[0])
            total = loss_tgt + loss_kd if total is None else total + loss_tgt + loss_kd
            detail.update(match_tok_gt=m_tg, match_tok_kd=m_kd, terms_tok_gt=t_tg, terms_tok_kd=t_kd,
                          targets_pred=(pboxes, pcount, pweight), device_counts=scal[1:3])
        losses["loss_total"] = total
        return losses, detail

    # ------------------------------------------------------------------ reference entry points
    def _encode_inputs(self, x_mm, cls_feat, text_feat):
        """Reference-layout inputs ([B,C,h,w] fp32, [B,C], [B,T,C]) -> modality-major fp32 rows (+ their 16-bit copy)."""
        B, C, h, w = x_mm.shape
        vis = torch.cat([cls_feat[:, None, :], x_mm.flatten(2).transpose(1, 2)], 1).reshape(B * (h * w + 1), C)
        out = torch.cat([vis, text_feat.reshape(-1, C)], 0).float().contiguous()
        out.lp = ops.cast_lp(out.detach())
        return out, B, h * w + 1, text_feat.shape[1]

    def forward_train(self, x_mm, img_metas, cls_feat=None, text_feat=None, gt_bbox=None, text_mask=None):
        enc_out, B, Nv, T = self._encode_inputs(x_mm, cls_feat, text_feat)
        output = self.forward_fused(enc_out, B, Nv, T, img_metas, text_mask)
        losses, _ = self.loss(output, gt_bbox, img_metas)
        return losses, output

    def forward_test(self, x_mm, img_metas, text_feat=None, cls_feat=None, with_bbox=False, with_mask=False, text_mask=None):
        enc_out, B, Nv, T = self._encode_inputs(x_mm, cls_feat, text_feat)
        return self.forward_fused(enc_out, B, Nv, T, img_metas, text_mask)

    def inference(self, box_cls, box_pred, image_sizes, wh=None):
        """head.inference (:577-604): softmax, drop the no-object column, cxcywh -> xyxy * (w, h).
        Returns (scores [B,nq], labels [B,nq], boxes_xyxy [B,nq,4]) instead of detectron2 Instances.
        `wh`: optional precomputed [B,4] device tensor of (w, h, w, h) per image (then `image_sizes` is not read)."""
        scores, labels = F.softmax(box_cls, dim=-1)[:, :, :-1].max(-1)
        cx, cy, w, h = box_pred.unbind(-1)
        xyxy = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)
        if wh is None:
            wh = torch.tensor([[s[1], s[0], s[1], s[0]] for s in image_sizes], dtype=xyxy.dtype, device=xyxy.device)
        return scores, labels, xyxy * wh[:, None, :]
