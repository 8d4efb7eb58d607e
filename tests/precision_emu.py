"""CPU emulation of the HIP engine's operand roundings on top of the oracle (TEST INFRASTRUCTURE).

The HIP encoder keeps the fp32 residual stream / LayerNorm / softmax / GELU / accumulators of the reference and rounds
only (a) the operands of every MFMA contraction and (b) the activations it stores for them (h, qkv, P, o, o2, h2, u, g2,
final output).  This module replays the oracle's encoder with a quantiser `q` applied at exactly those points, so the
error budget of a storage format (bf16: 8 significand bits, fp16: 11, "split" = hi+lo pair of bf16 ~ 16 bits) can be
measured against the reference fixtures WITHOUT a GPU, per stage (which roundings are switched on).

Used by tests/test_precision_cpu.py and as a script:  python -m tests.precision_emu [fixture ...]
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import simvg_cpu as O, weights as W   # noqa: E402

STAGES = ("w", "h", "qkv", "p", "o", "o2", "h2", "u", "g2", "out", "patch")


def quantiser(fmt):
    if fmt == "fp32":
        return lambda t: t
    if fmt == "bf16":
        return lambda t: t.to(torch.bfloat16).float()
    if fmt == "fp16":
        return lambda t: t.to(torch.float16).float()
    if fmt == "split":     # bf16 hi + bf16 lo (what a 3-product split GEMM sees)
        def q(t):
            hi = t.to(torch.bfloat16).float()
            return hi + (t - hi).to(torch.bfloat16).float()
        return q
    raise ValueError(fmt)


class Emu:
    """fmt: storage format; on: set of STAGES whose rounding is active (default all)."""

    def __init__(self, fmt="bf16", on=STAGES, fmt_over=None):
        self.q0 = quantiser(fmt)
        self.on = set(on)
        self.over = {k: quantiser(v) for k, v in (fmt_over or {}).items()}

    def q(self, stage, t):
        if stage not in self.on:
            return t
        return self.over.get(stage, self.q0)(t)

    def lin(self, sd, key, x, split, wstage="w"):
        def f(e):
            return lambda t: F.linear(t, self.q(wstage, sd[f"{key}.{e}.weight"]), sd[f"{key}.{e}.bias"])
        return O._mw(x, split, f("A"), f("B"))

    def layer(self, sd, cfg, x, mask, split, i, p="vis_enc.beit3."):
        L = f"{p}encoder.layers.{i}."
        B, N, D = x.shape
        H, d, eps = cfg.heads, D // cfg.heads, cfg.ln_eps
        h = self.q("h", O._mw_ln(sd, L + "self_attn_layer_norm", x, split, eps))
        qq = self.q("qkv", self.lin(sd, L + "self_attn.q_proj", h, split))
        kk = self.q("qkv", self.lin(sd, L + "self_attn.k_proj", h, split))
        vv = self.q("qkv", self.lin(sd, L + "self_attn.v_proj", h, split))
        qq = qq.view(B, N, H, d).transpose(1, 2)
        kk = kk.view(B, N, H, d).transpose(1, 2)
        vv = vv.view(B, N, H, d).transpose(1, 2)
        w = (qq @ kk.transpose(-1, -2)) * (d ** -0.5)          # the kernel folds the scale into the fp32 score
        w = w.masked_fill(mask[:, None, None, :].to(torch.bool), float("-inf"))
        w = F.softmax(w, dim=-1, dtype=torch.float32)
        # kernel: P = exp2(s - max) rounded as MFMA operand, the 1/l normalisation applied to the fp32 accumulator
        l = w.max(dim=-1, keepdim=True)[0]
        pn = self.q("p", w / l)
        a = ((pn @ vv) * l).transpose(1, 2).reshape(B, N, D)
        a = self.q("o", a)
        a = self.q("o2", O._mw_ln(sd, L + "self_attn.inner_attn_ln", a, split, eps))
        x = x + self.lin(sd, L + "self_attn.out_proj", a, split)
        h = self.q("h2", O._mw_ln(sd, L + "final_layer_norm", x, split, eps))

        def ffn(t, e):
            u = self.q("u", F.linear(t, self.q("w", sd[f"{L}ffn.{e}.fc1.weight"]), sd[f"{L}ffn.{e}.fc1.bias"]))
            g = F.gelu(u)
            g = F.layer_norm(g, g.shape[-1:], sd[f"{L}ffn.{e}.ffn_layernorm.weight"], sd[f"{L}ffn.{e}.ffn_layernorm.bias"], eps)
            return F.linear(self.q("g2", g), self.q("w", sd[f"{L}ffn.{e}.fc2.weight"]), sd[f"{L}ffn.{e}.fc2.bias"])

        return x + O._mw(h, split, lambda t: ffn(t, "A"), lambda t: ffn(t, "B"))

    def encoder(self, sd, cfg, img, ids, pad, p="vis_enc.beit3."):
        P = cfg.patch_size
        x1 = F.conv2d(self.q("patch", img), self.q("patch", sd[p + "vision_embed.proj.weight"]),
                      sd[p + "vision_embed.proj.bias"], stride=P)
        x1 = x1.flatten(2).transpose(1, 2)
        B = x1.shape[0]
        x1 = torch.cat([sd[p + "vision_embed.cls_token"].expand(B, -1, -1), x1], dim=1)
        split = x1.shape[1]
        x2 = F.embedding(ids, sd[p + "text_embed.weight"])
        x = torch.cat([x1, x2], dim=1)
        T = x2.shape[1]
        mask = torch.cat([torch.zeros(x1.shape[:-1]).bool(), pad.bool()], dim=1)
        pos = torch.cat([sd[p + "encoder.embed_positions.A.weight"][2:2 + split],
                         sd[p + "encoder.embed_positions.B.weight"][2:2 + T]], dim=0)[None]
        x = (x + pos) * (1 - mask.unsqueeze(-1).type_as(x))
        for i in range(cfg.layers):
            x = self.layer(sd, cfg, x, mask, split, i, p)
        x = self.q("out", O._mw_ln(sd, p + "encoder.layer_norm", x, split, cfg.ln_eps))
        return x[:, 1:-T], x[:, -T:], x[:, 0]


@torch.no_grad()
def box_errors(fx, emu):
    """-> (decoder L1, token L1, cls feature abs error / max) of the emulated engine vs the reference fixture"""
    cfg = O.make_cfg(fx["vit"], fx["num_queries"], fx["img_size"])
    sd = W.reference_init_state_dict(cfg, fx["wseed"]) if fx.get("refinit") else W.golden_state_dict(cfg, fx["wseed"])
    batch = W.synthetic_batch(cfg, fx["B"], fx["iseed"], fx["grec"])
    metas = [dict(m) for m in batch["img_metas"]]
    for m in metas:
        m["batch_input_shape"] = tuple(batch["img"].shape[-2:])
    img_feat, text_feat, cls_feat = emu.encoder(sd, cfg, batch["img"], batch["ref_expr_inds"], batch["text_attention_mask"])
    out = O.head_forward_general(sd, cfg, img_feat, text_feat, cls_feat, batch["text_attention_mask"], metas)
    dec = float((out["dec_boxes"] - fx["dec_boxes"]).abs().sum(-1).max())
    tok = float((out["tok_boxes"] - fx["tok_boxes"]).abs().sum(-1).max())
    cls = float((cls_feat - fx["cls_feat"]).abs().max() / fx["cls_feat"].abs().max())
    return dec, tok, cls


def main():
    torch.set_num_threads(os.cpu_count())
    names = sys.argv[1:] or ["tiny_nq1", "tiny_nq10_grec"]
    gold = os.path.join(ROOT, "tests", "golden")
    for name in names:
        fx = torch.load(os.path.join(gold, name + ".pt"), weights_only=False)
        rows = [("fp32", Emu("fp32")), ("bf16 all", Emu("bf16")), ("fp16 all", Emu("fp16")), ("split all", Emu("split")),
                ("bf16, fp32 out", Emu("bf16", [s for s in STAGES if s != "out"]))]
        for s in STAGES:
            rows.append((f"bf16 only {s}", Emu("bf16", [s])))
        for label, emu in rows:
            dec, tok, cls = box_errors(fx, emu)
            print(f"{name:24s} {label:18s} dec L1 {dec:.2e}  tok L1 {tok:.2e}  cls rel {cls:.2e}", flush=True)


if __name__ == "__main__":
    main()
