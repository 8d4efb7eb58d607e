"""GPU: the decoder layers of the head as few launches (csrc/decoder.hip) against plain PyTorch -- detrex BaseTransformerLayer
as the reference configures it (transformer.py:93-131: self_attn, norm, cross_attn, norm, ffn, norm; nn.MultiheadAttention with
8 heads, key_pos added to the keys only)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
DEV = "cuda"
E, H = 256, 8


def _ops():
    from simvg_amd import hip_ops
    return hip_ops


def _layer_params(g, ffn=2048, scale=1.0):
    r = lambda *s: torch.randn(*s, generator=g)
    W = dict(Ws=r(3 * E, E) * E ** -0.5 * scale, bs=r(3 * E) * 0.1, Wso=r(E, E) * E ** -0.5 * scale, bso=r(E) * 0.1,
             g0=1 + 0.1 * r(E), b0=0.1 * r(E), Wc=r(3 * E, E) * E ** -0.5 * scale, bc=r(3 * E) * 0.1,
             Wco=r(E, E) * E ** -0.5 * scale, bco=r(E) * 0.1, g1=1 + 0.1 * r(E), b1=0.1 * r(E),
             W1=r(ffn, E) * E ** -0.5, b1f=r(ffn) * 0.1, W2=r(E, ffn) * ffn ** -0.5, b2f=r(E) * 0.1, g2=1 + 0.1 * r(E), b2=0.1 * r(E))
    return W


ATTN_KEYS = ("Ws", "bs", "Wso", "bso", "g0", "b0", "Wc", "bc", "Wco", "bco", "g1", "b1")


def _mha(q_in, k_in, v_in, Wi, bi, Wo, bo, B, Lq, Lk, kpm=None, dm=None):
    """nn.MultiheadAttention (batch-major rows [B*L, E]), dropout on the probabilities as explicit multipliers"""
    q = (q_in @ Wi[:E].t() + bi[:E]).view(B, Lq, H, E // H).transpose(1, 2)
    k = (k_in @ Wi[E:2 * E].t() + bi[E:2 * E]).view(B, Lk, H, E // H).transpose(1, 2)
    v = (v_in @ Wi[2 * E:].t() + bi[2 * E:]).view(B, Lk, H, E // H).transpose(1, 2)
    s = q @ k.transpose(-1, -2) * (E // H) ** -0.5
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :].bool(), float("-inf"))
    p = s.softmax(-1)
    pd = p * dm if dm is not None else p
    o = (pd @ v).transpose(1, 2).reshape(B * Lq, E)
    return o @ Wo.t() + bo, p


def _attn_block_ref(tgt, qpos, W, src_k, src_v, B, R, Lk, kpm, dm0, dm1):
    a, p0 = _mha(tgt + qpos, tgt + qpos, tgt, W["Ws"], W["bs"], W["Wso"], W["bso"], B, R, R, dm=dm0)
    t1 = F.layer_norm(tgt + a, (E,), W["g0"], W["b0"], 1e-5)
    c, p1 = _mha(t1 + qpos, src_k, src_v, W["Wc"], W["bc"], W["Wco"], W["bco"], B, R, Lk, kpm=kpm, dm=dm1)
    t2 = F.layer_norm(t1 + c, (E,), W["g1"], W["b1"], 1e-5)
    return t1, t2, p0, p1


@pytest.mark.parametrize("B,R,Lk,kind,drop", [(5, 1, 400, "mem16", False), (3, 10, 400, "mem16", True), (4, 3, 20, "text", True),
                                              (2, 16, 37, "mem32", False), (64, 1, 400, "mem16", True), (2, 6, 400, "mem16_pos_per_sample", False)])
def test_attention_block_forward(B, R, Lk, kind, drop):
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + R * 10 + Lk)
    W = _layer_params(g)
    M = B * R
    tgt, qpos = torch.randn(M, E, generator=g), torch.randn(M, E, generator=g)
    kv_rows, kv_off = (Lk + 1, 1) if kind.startswith("mem") else (Lk, 0)
    src = torch.randn(B * kv_rows, E, generator=g)
    if kind.startswith("mem16"):
        src = src.to(ops.LP()).float()                      # the memory rows exist in the 16-bit storage format
    kpos = torch.randn(B * Lk if kind.endswith("per_sample") else Lk, E, generator=g)
    kpm = (torch.rand(B, Lk, generator=g) < 0.3).to(torch.uint8)
    kpm[:, 0] = 0
    dm0 = ((torch.rand(B, H, R, R, generator=g) > 0.1).float() / 0.9) if drop else None
    dm1 = ((torch.rand(B, H, R, Lk, generator=g) > 0.1).float() / 0.9) if drop else None
    rows = src.view(B, kv_rows, E)[:, kv_off:kv_off + Lk]
    keys = (rows + kpos.view(-1, Lk, E)).reshape(B * Lk, E).double()
    vals = rows.reshape(B * Lk, E).double()
    Wd = {k: v.double() for k, v in W.items()}
    t1, t2, p0, p1 = _attn_block_ref(tgt.double(), qpos.double(), Wd, keys, vals, B, R, Lk, kpm,
                                     None if dm0 is None else dm0.double(), None if dm1 is None else dm1.double())
    d = lambda t: None if t is None else t.to(DEV)
    srcd = src.to(DEV).to(ops.LP()) if kind.startswith("mem16") else src.to(DEV)
    out = ops.dec_attn_fwd(d(tgt), d(qpos), [W[k].to(DEV) for k in ATTN_KEYS], srcd, B, R, Lk, kv_rows=kv_rows, kv_off=kv_off,
                           kpos=d(kpos), kpm=d(kpm), dm0=d(dm0), dm1=d(dm1))
    torch.cuda.synchronize()
    for name, ref in (("P0", p0), ("t1", t1), ("P1", p1), ("t2", t2)):
        err = float((out[name].double().cpu() - ref).abs().max())
        print(f"[attn block fwd B={B} R={R} Lk={Lk} {kind}] {name}: max abs err {err:.2e}")
        assert err <= 2e-5, (name, err)


@pytest.mark.parametrize("B,R,Lk,kind,drop,slabs", [(5, 1, 400, "mem16", False, 0), (3, 10, 400, "mem16", True, 3), (4, 3, 20, "text", True, 0),
                                                    (2, 16, 37, "mem32", False, 2), (2, 5, 400, "mem16_pos_per_sample", True, 0)])
def test_attention_block_backward(B, R, Lk, kind, drop, slabs):
    """d tgt, d qpos, d source rows and all 12 parameter gradients of the block against autograd of the plain PyTorch layer (double
    precision); the gradient of the block's output arrives as a tensor and / or as slabs that the kernel sums."""
    ops = _ops()
    g = torch.Generator().manual_seed(7 + B * 1000 + R * 10 + Lk)
    W = _layer_params(g)
    M = B * R
    tgt, qpos = torch.randn(M, E, generator=g), torch.randn(M, E, generator=g)
    kv_rows, kv_off = (Lk + 1, 1) if kind.startswith("mem") else (Lk, 0)
    src = torch.randn(B * kv_rows, E, generator=g)
    if kind.startswith("mem16"):
        src = src.to(ops.LP()).float()
    kpos = torch.randn(B * Lk if kind.endswith("per_sample") else Lk, E, generator=g)
    kpm = (torch.rand(B, Lk, generator=g) < 0.3).to(torch.uint8)
    kpm[:, 0] = 0
    dm0 = ((torch.rand(B, H, R, R, generator=g) > 0.1).float() / 0.9) if drop else None
    dm1 = ((torch.rand(B, H, R, Lk, generator=g) > 0.1).float() / 0.9) if drop else None
    dt2 = torch.randn(M, E, generator=g)
    dsl = torch.randn(slabs, M, E, generator=g) if slabs else None
    dy = dt2 + (dsl.sum(0) if slabs else 0)
    # reference: autograd in double precision
    Wd = {k: v.double().requires_grad_(True) for k, v in W.items() if k in ATTN_KEYS}
    tg, qp, sr = tgt.double().requires_grad_(True), qpos.double().requires_grad_(True), src.double().requires_grad_(True)
    rows = sr.view(B, kv_rows, E)[:, kv_off:kv_off + Lk]
    keys = (rows + kpos.double().view(-1, Lk, E)).reshape(B * Lk, E)
    vals = rows.reshape(B * Lk, E)
    t1, t2, p0, p1 = _attn_block_ref(tg, qp, Wd, keys, vals, B, R, Lk, kpm, None if dm0 is None else dm0.double(),
                                     None if dm1 is None else dm1.double())
    t2.backward(dy.double())
    d = lambda t: None if t is None else t.to(DEV)
    srcd = src.to(DEV).to(ops.LP()) if kind.startswith("mem16") else src.to(DEV)
    Wl = [W[k].to(DEV) for k in ATTN_KEYS]
    tgd, qpd = d(tgt), d(qpos)
    kw = dict(kv_rows=kv_rows, kv_off=kv_off, kpos=d(kpos), dm0=d(dm0), dm1=d(dm1))
    saved = ops.dec_attn_fwd(tgd, qpd, Wl, srcd, B, R, Lk, kpm=d(kpm), **kw)
    outs = []
    for acc in (False, True):            # d(source rows) written into garbage == added to zeros + the same again
        dsrc = torch.full((B * kv_rows, E), float("nan") if not acc else 1.0, device=DEV)
        d_tgt, d_qpos, grads = ops.dec_attn_bwd(saved, tgd, qpd, Wl, srcd, B, R, Lk, dt2=d(dt2), dt2_slabs=d(dsl), dsrc=dsrc,
                                                dsrc_accumulate=acc, **kw)
        torch.cuda.synchronize()
        outs.append((d_tgt.clone(), d_qpos.clone(), [x.clone() for x in grads], dsrc.clone()))
    d_tgt, d_qpos, grads, dsrc = outs[0]
    assert torch.equal(outs[1][3], dsrc + 1.0) or float((outs[1][3] - 1.0 - dsrc).abs().max()) <= 1e-6 * float(dsrc.abs().max())
    for x, y in zip(outs[0][2], outs[1][2]):
        assert torch.equal(x, y)          # parameter gradients: bit-reproducible
    def rel(got, ref):
        return float((got.double().cpu() - ref).abs().max() / max(float(ref.abs().max()), 1e-12))
    checks = [("d_tgt", d_tgt, tg.grad), ("d_qpos", d_qpos, qp.grad), ("d_src", dsrc, sr.grad)]
    checks += [("d" + k, gr, Wd[k].grad) for k, gr in zip(ATTN_KEYS, grads)]
    worst = 0.0
    for name, got, ref in checks:
        if name == "dbc":                 # the key third of the in-projection bias has an exactly-zero gradient (rounding noise in autograd)
            assert float(got[E:2 * E].abs().max()) == 0.0
            e = max(rel(got[:E], ref[:E]), rel(got[2 * E:], ref[2 * E:]))
        else:
            e = rel(got, ref)
        worst = max(worst, e)
        assert e <= 2e-5, (name, e)
    print(f"[attn block bwd B={B} R={R} Lk={Lk} {kind} drop={drop} slabs={slabs}] worst relative error over {len(checks)} gradients {worst:.2e}")


@pytest.mark.parametrize("M,Fd,post,drop", [(64, 2048, True, True), (5, 512, False, False), (640, 2048, True, False), (37, 512, True, True)])
def test_ffn_block_forward_backward(M, Fd, post, drop):
    """FFN + norm (+ post-norm) of a decoder layer, hidden-split kernels, against autograd in double precision; the gradient of t2
    is d_r3 + the sum of the slabs (what the attention block's backward forms)."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + Fd)
    W = _layer_params(g, ffn=Fd)
    gP, bP = 1 + 0.1 * torch.randn(E, generator=g), 0.1 * torch.randn(E, generator=g)
    t2 = torch.randn(M, E, generator=g)
    m1 = ((torch.rand(M, Fd, generator=g) > 0.1).float() / 0.9) if drop else None
    m2 = ((torch.rand(M, E, generator=g) > 0.1).float() / 0.9) if drop else None
    d_t3 = torch.randn(M, E, generator=g)
    d_hs = torch.randn(M, E, generator=g) if post else None
    P = {k: W[k].double().requires_grad_(True) for k in ("W1", "b1f", "W2", "b2f", "g2", "b2")}
    gPd, bPd = gP.double().requires_grad_(True), bP.double().requires_grad_(True)
    x = t2.double().requires_grad_(True)
    h = F.relu(x @ P["W1"].t() + P["b1f"])
    if drop:
        h = h * m1.double()
    y = h @ P["W2"].t() + P["b2f"]
    if drop:
        y = y * m2.double()
    t3 = F.layer_norm(x + y, (E,), P["g2"], P["b2"], 1e-5)
    loss = (t3 * d_t3.double()).sum()
    if post:
        hs = F.layer_norm(t3, (E,), gPd, bPd, 1e-5)
        loss = loss + (hs * d_hs.double()).sum()
    loss.backward()
    d = lambda t: None if t is None else t.to(DEV)
    Wd = {k: v.to(DEV) for k, v in W.items()}
    t2d = d(t2)
    sv = ops.dec_ffn_fwd(t2d, Wd["W1"], Wd["b1f"], Wd["W2"], Wd["b2f"], Wd["g2"], Wd["b2"], gP=d(gP) if post else None,
                         bP=d(bP) if post else None, m1=d(m1), m2=d(m2))
    torch.cuda.synchronize()
    rel = lambda got, ref: float((got.double().cpu() - ref).abs().max() / max(float(ref.abs().max()), 1e-12))
    assert rel(sv["t3"], t3.detach()) <= 5e-6
    if post:
        assert rel(sv["hs"], hs.detach()) <= 5e-6
    outs = []
    for _ in range(2):
        d_r3, slabs, grads = ops.dec_ffn_bwd(sv, t2d, Wd["W1"], Wd["W2"], Wd["g2"], gP=d(gP) if post else None, d_t3=d(d_t3), d_hs=d(d_hs),
                                             m1=d(m1), m2=d(m2))
        torch.cuda.synchronize()
        outs.append([d_r3.clone(), slabs.clone()] + [None if x_ is None else x_.clone() for x_ in grads])
    for x_, y_ in zip(*outs):
        assert (x_ is None and y_ is None) or torch.equal(x_, y_)          # bit-reproducible
    d_r3, slabs, dW1, db1, dW2, db2, dg2, db2n, dgP, dbP = outs[0]
    checks = [("d_t2", d_r3 + slabs.sum(0), x.grad), ("dW1", dW1, P["W1"].grad), ("db1", db1, P["b1f"].grad), ("dW2", dW2, P["W2"].grad),
              ("db2", db2, P["b2f"].grad), ("dg2", dg2, P["g2"].grad), ("db2n", db2n, P["b2"].grad)]
    if post:
        checks += [("dgP", dgP, gPd.grad), ("dbP", dbP, bPd.grad)]
    worst = 0.0
    for name, got, ref in checks:
        e = rel(got, ref)
        worst = max(worst, e)
        assert e <= 2e-5, (name, e)
    print(f"[ffn block M={M} Fd={Fd} post={post} drop={drop}] forward <= 5e-6, worst relative gradient error {worst:.2e}")


@pytest.mark.parametrize("M,lead", [(64, True), (192, False), (7, True)])
def test_prediction_heads_node(M, lead):
    """class Linear + 3-layer box MLP + sigmoid (+ the token branch's leading Linear) as one autograd node (PredHeadFn) against
    plain PyTorch: outputs and every gradient, with gradients arriving for the logits, the boxes and (lead) the leading output."""
    from simvg_amd.models.heads.functions import PredHeadFn
    g = torch.Generator().manual_seed(M)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    x = r(M, E)
    P = [r(E, E, sc=E ** -0.5), r(E, sc=0.1)] if lead else [None, None]
    P += [r(2, E, sc=E ** -0.5), r(2, sc=0.1), r(E, E, sc=E ** -0.5), r(E, sc=0.1), r(E, E, sc=E ** -0.5), r(E, sc=0.1), r(4, E, sc=E ** -0.5), r(4, sc=0.1)]
    dl, db_, dxm = r(M, 2), r(M, 4), r(M, E)
    leaves = lambda dev, dt: [None if t is None else t.clone().to(dev).to(dt).requires_grad_(True) for t in [x] + P]
    c = leaves("cpu", torch.float64)
    xm = c[0] @ c[1].t() + c[2] if lead else c[0]
    lg = xm @ c[3].t() + c[4]
    bx = torch.sigmoid(F.relu(F.relu(xm @ c[5].t() + c[6]) @ c[7].t() + c[8]) @ c[9].t() + c[10])
    loss = (lg * dl.double()).sum() + (bx * db_.double()).sum() + ((xm * dxm.double()).sum() if lead else 0)
    loss.backward()
    d = leaves(DEV, torch.float32)
    lgd, bxd, xmd = PredHeadFn.apply(*d)
    outs, grads = [lgd, bxd] + ([xmd] if lead else []), [dl.to(DEV), db_.to(DEV)] + ([dxm.to(DEV)] if lead else [])
    torch.autograd.backward(outs, grads)
    rel = lambda got, ref: float((got.double().cpu() - ref).abs().max() / max(float(ref.abs().max()), 1e-12))
    assert rel(lgd, lg.detach()) <= 5e-6 and rel(bxd, bx.detach()) <= 5e-6
    assert (xmd is None) == (not lead)
    for i, (a, b) in enumerate(zip(d, c)):
        if b is not None:
            assert rel(a.grad, b.grad) <= 2e-5, (i, rel(a.grad, b.grad))
