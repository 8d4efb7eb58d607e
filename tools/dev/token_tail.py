"""Which rounding points of the 16-bit engine set the TAIL of the token-branch box error at full batch (dev container, CPU).

Replays the oracle's encoder with the engine's rounding points (tests/precision_emu.py) on the full-size fixture's batch
(`base_nq1_full`: ViT-B, 64 pairs, harsh weights) and reports mean / p99 / max L1 of the token boxes and of the decoder boxes
against the REFERENCE's (the fixture), for
  * every single stage alone (which rounding matters),
  * all stages with the rows the token branch reads (CLS + the 20 text rows of each sample) EXEMPT from chosen sets of
    roundings (what a higher-precision path for those 5 % of the rows would buy).

    python tools/dev/token_tail.py [fixture] [variant ...]
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import simvg_cpu as O, weights as W          # noqa: E402
from tests.precision_emu import STAGES, quantiser        # noqa: E402


class RowEmu:
    """fmt at every stage in `on`; for stages in `exempt` the rows in `rows` (indices into the 421 tokens) keep `hi_fmt`.
    Weight stages: "w" = expert A (vision rows, CLS included), "wB" = expert B (text rows)."""

    def __init__(self, fmt="fp16", on=STAGES + ("wB",), exempt=(), rows=(), hi_fmt="fp32"):
        self.qf, self.qh = quantiser(fmt), quantiser(hi_fmt)
        self.on, self.exempt, self.rows = set(on), set(exempt), list(rows)

    def q(self, stage, t, rowdim=1):
        if stage not in self.on:
            return t
        out = self.qf(t)
        if stage in self.exempt and self.rows:
            idx = torch.tensor(self.rows)
            out = out.index_copy(rowdim, idx, self.qh(t.index_select(rowdim, idx)))
        return out

    def w(self, e, t, which=""):
        stage = "w" if e == "A" else "wB"
        if stage not in self.on:
            return t
        if stage in self.exempt or (which and (which + ("" if e == "A" else "B")) in self.exempt):
            return self.qh(t)
        if self.layer_lo is not None and not (self.layer_lo <= self.cur_layer < self.layer_hi):
            return self.qh(t)
        return self.qf(t)

    layer_lo = None
    cur_layer = 0

    def lin(self, sd, key, x, split):
        which = "wout" if key.endswith("out_proj") else "wqkv"
        def f(e):
            return lambda t: F.linear(t, self.w(e, sd[f"{key}.{e}.weight"], which), sd[f"{key}.{e}.bias"])
        return O._mw(x, split, f("A"), f("B"))

    def layer(self, sd, cfg, x, mask, split, i, p="vis_enc.beit3."):
        L = f"{p}encoder.layers.{i}."
        B, N, D = x.shape
        H, d, eps = cfg.heads, D // cfg.heads, cfg.ln_eps
        h = self.q("h", O._mw_ln(sd, L + "self_attn_layer_norm", x, split, eps))
        qq = self.q("q", self.lin(sd, L + "self_attn.q_proj", h, split))
        kk = self.q("kv", self.lin(sd, L + "self_attn.k_proj", h, split))
        vv = self.q("kv", self.lin(sd, L + "self_attn.v_proj", h, split))
        qq = qq.view(B, N, H, d).transpose(1, 2)
        kk = kk.view(B, N, H, d).transpose(1, 2)
        vv = vv.view(B, N, H, d).transpose(1, 2)
        w = (qq @ kk.transpose(-1, -2)) * (d ** -0.5)
        w = w.masked_fill(mask[:, None, None, :].to(torch.bool), float("-inf"))
        w = F.softmax(w, dim=-1, dtype=torch.float32)
        l = w.max(dim=-1, keepdim=True)[0]
        pn = self.q("p", w / l, rowdim=2)
        a = ((pn @ vv) * l).transpose(1, 2).reshape(B, N, D)
        a = self.q("o", a)
        a = self.q("o2", O._mw_ln(sd, L + "self_attn.inner_attn_ln", a, split, eps))
        x = x + self.lin(sd, L + "self_attn.out_proj", a, split)
        h = self.q("h2", O._mw_ln(sd, L + "final_layer_norm", x, split, eps))

        def ffn(t, e, rows_here):
            u = F.linear(t, self.w(e, sd[f"{L}ffn.{e}.fc1.weight"], "w1"), sd[f"{L}ffn.{e}.fc1.bias"])
            u = self._q_part("u", u, rows_here)
            g = F.gelu(u)
            g = F.layer_norm(g, g.shape[-1:], sd[f"{L}ffn.{e}.ffn_layernorm.weight"], sd[f"{L}ffn.{e}.ffn_layernorm.bias"], eps)
            return F.linear(self._q_part("g2", g, rows_here), self.w(e, sd[f"{L}ffn.{e}.fc2.weight"], "w2"), sd[f"{L}ffn.{e}.fc2.bias"])

        ra = [r for r in self.rows if r < split]
        rb = [r - split for r in self.rows if r >= split]
        return x + O._mw(h, split, lambda t: ffn(t, "A", ra), lambda t: ffn(t, "B", rb))

    def _q_part(self, stage, t, rows):
        if stage not in self.on:
            return t
        out = self.qf(t)
        if stage in self.exempt and rows:
            idx = torch.tensor(rows)
            out = out.index_copy(1, idx, self.qh(t.index_select(1, idx)))
        return out

    def encoder(self, sd, cfg, img, ids, pad, p="vis_enc.beit3."):
        P = cfg.patch_size
        qp = (lambda t: self.qf(t)) if "patch" in self.on else (lambda t: t)
        x1 = F.conv2d(qp(img), qp(sd[p + "vision_embed.proj.weight"]), sd[p + "vision_embed.proj.bias"], stride=P)
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
            self.cur_layer = i
            x = self.layer(sd, cfg, x, mask, split, i, p)
        x = O._mw_ln(sd, p + "encoder.layer_norm", x, split, cfg.ln_eps)
        xq = self.qf(x) if "out" in self.on else x          # the decoder's memory is the 16-bit copy, the token branch reads fp32
        return xq[:, 1:-T], x[:, -T:], x[:, 0]


def stats(a, b):
    d = (a - b).abs().sum(-1).reshape(-1).double()
    return float(d.mean()), float(torch.quantile(d, 0.99)), float(d.max())


@torch.no_grad()
def run(fx, emu, cache):
    if "sd" not in cache:
        cfg = O.make_cfg(fx["vit"], fx["num_queries"], fx["img_size"])
        cache["cfg"] = cfg
        cache["sd"] = W.golden_state_dict(cfg, fx["wseed"])
        cache["batch"] = W.synthetic_batch(cfg, fx["B"], fx["iseed"], fx["grec"])
    cfg, sd, batch = cache["cfg"], cache["sd"], cache["batch"]
    metas = [dict(m) for m in batch["img_metas"]]
    for m in metas:
        m["batch_input_shape"] = tuple(batch["img"].shape[-2:])
    nb = int(os.environ.get("TAIL_CHUNK", "16"))
    outs = []
    for s in range(0, fx["B"], nb):              # batch-independent (eval mode): chunks bound the host memory
        sl = slice(s, s + nb)
        f = emu.encoder(sd, cfg, batch["img"][sl], batch["ref_expr_inds"][sl], batch["text_attention_mask"][sl])
        o = O.head_forward_general(sd, cfg, *f, batch["text_attention_mask"][sl], metas[sl])
        outs.append((o["dec_boxes"], o["tok_boxes"]))
    dec = torch.cat([o[0] for o in outs], dim=1)
    tok = torch.cat([o[1] for o in outs], dim=1)
    return stats(dec, fx["dec_boxes"]), stats(tok, fx["tok_boxes"])


def main():
    torch.set_num_threads(os.cpu_count())
    name = sys.argv[1] if len(sys.argv) > 1 else "base_nq1_full"
    fx = torch.load(os.path.join(ROOT, "tests", "golden", name + ".pt"), weights_only=False)
    Nv = (fx["img_size"] // 32) ** 2 + 1
    tok_rows = [0] + list(range(Nv, Nv + 20))
    ALL = STAGES + ("wB", "q", "kv")
    ALL = tuple(s for s in ALL if s != "qkv")
    own = ("h", "o2", "h2", "u", "g2")
    variants = {
        "fp32": RowEmu("fp32"),
        "fp16 all": RowEmu("fp16", ALL),
        "exempt own-row GEMM operands (h o2 h2 u g2), text+CLS": RowEmu("fp16", ALL, own, tok_rows),
        "  + weights of expert B": RowEmu("fp16", ALL, own + ("wB",), tok_rows),
        "  + weights of both experts": RowEmu("fp16", ALL, own + ("wB", "w"), tok_rows),
        "  + q o (own rows of the attention)": RowEmu("fp16", ALL, own + ("wB", "q", "o"), tok_rows),
        "  + q o p": RowEmu("fp16", ALL, own + ("wB", "q", "o", "p"), tok_rows),
        "exempt weights only (both experts)": RowEmu("fp16", ALL, ("w", "wB"), tok_rows),
        "exempt own-row incl. q o p, fp16 weights": RowEmu("fp16", ALL, own + ("q", "o", "p"), tok_rows),
    }
    for m in ("wqkv", "wout", "w1", "w2"):
        variants[f"weights exact: {m} (expert A)"] = RowEmu("fp16", ALL, (m,), tok_rows)
    variants["weights exact: w1 w2 (both experts)"] = RowEmu("fp16", ALL, ("w1", "w2", "w1B", "w2B"), tok_rows)
    variants["weights exact: wqkv wout (both experts)"] = RowEmu("fp16", ALL, ("wqkv", "wout", "wqkvB", "woutB"), tok_rows)
    for lo, hi in ((0, 6), (6, 12), (9, 12)):
        e = RowEmu("fp16", ALL, (), tok_rows)
        e.layer_lo, e.layer_hi = lo, hi
        variants[f"weights fp16 only in layers [{lo},{hi}), exact elsewhere"] = e
    for s in ALL:
        variants[f"only {s}"] = RowEmu("fp16", (s,))
    want = sys.argv[2:]
    cache = {}
    for label, emu in variants.items():
        if want and not any(w in label for w in want):
            continue
        (dm, dp, dx), (tm, tp, tx) = run(fx, emu, cache)
        print(f"{label:60s} dec mean {dm:.2e} p99 {dp:.2e} max {dx:.2e} | tok mean {tm:.2e} p99 {tp:.2e} max {tx:.2e}", flush=True)


if __name__ == "__main__":
    main()
