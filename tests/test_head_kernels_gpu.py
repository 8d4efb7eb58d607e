"""GPU op-level parity for the decoder-head kernels: exact-fp32 MFMA GEMM, small attention, on-device
Hungarian matcher (against SciPy, which the reference uses through detrex), criterion value + gradients
(against autograd through the oracle's restatement of SetCriterion)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(got, ref, tol, what=""):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(float(ref.abs().max()), 1e-6)
    err = float((got - ref).abs().max())
    assert err <= tol * scale, f"{what}: err {err:.3g} scale {scale:.3g}"


@pytest.mark.parametrize("M,N,K,relu", [(64, 256, 768, False), (10, 2, 256, False), (640, 2048, 256, True), (1280, 256, 768, False),
                                        (64, 256, 2048, True), (5, 3, 13, False), (33, 70, 20, True), (2560, 256, 256, False)])
def test_linear_f32_fwd_bwd(M, N, K, relu):
    from simvg_amd.models.heads.functions import LinearF32
    g = torch.Generator().manual_seed(M + N)
    x, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * K ** -0.5, torch.randn(N, generator=g)
    xr, Wr, br = (t.clone().requires_grad_(True) for t in (x, W, b))
    y_ref = F.linear(xr, Wr, br)
    if relu:
        y_ref = F.relu(y_ref)
    dy = torch.randn(M, N, generator=g)
    y_ref.backward(dy)
    xd, Wd, bd = (t.clone().to(DEV).requires_grad_(True) for t in (x, W, b))
    y = LinearF32.apply(xd, Wd, bd, relu)
    y.backward(dy.to(DEV))
    close(y, y_ref, 1e-5, "y"); close(xd.grad, xr.grad, 1e-5, "dx"); close(Wd.grad, Wr.grad, 1e-5, "dW"); close(bd.grad, br.grad, 1e-5, "db")


def test_linear_f32_strided_unaligned_views():
    """operands that are column-offset views (base pointer not 16-B aligned, row stride != K) take the scalar-load path"""
    from simvg_amd import hip_ops as ops
    g = torch.Generator().manual_seed(5)
    big_a, big_b = torch.randn(40, 100, generator=g).to(DEV), torch.randn(24, 90, generator=g).to(DEV)
    A, Bw = big_a[:, 3:3 + 37], big_b[:, 1:1 + 37]          # [40,37], [24,37]
    out = torch.empty(40, 24, device=DEV)
    ops.gemm_f32(A, A.stride(0), 1, Bw, 1, Bw.stride(0), out, 40, 24, 37)
    close(out, A.cpu() @ Bw.cpu().t(), 1e-5, "strided NT")
    out2 = torch.empty(37, 24, device=DEV)                   # A^T (k strided) x B (n strided)
    Bk = big_b[:, 5:5 + 40].t()                              # B(k,n) = big_b[n, 5+k]
    ops.gemm_f32(A, 1, A.stride(0), Bk, 1, big_b.stride(0), out2, 37, 24, 40)
    close(out2, A.cpu().t() @ big_b[:, 5:45].cpu().t(), 1e-5, "strided TN")


@pytest.mark.parametrize("B,Lq,Lk,kv_off", [(3, 1, 400, True), (2, 10, 20, False), (2, 10, 10, False), (4, 7, 400, True)])
@pytest.mark.parametrize("drop", [False, True])
def test_small_attention_fwd_bwd(B, Lq, Lk, kv_off, drop):
    from simvg_amd.models.heads.functions import SmallAttention
    H, E = 8, 256
    g = torch.Generator().manual_seed(B * 100 + Lq + Lk)
    rows = Lk + 1 if kv_off else Lk
    q = torch.randn(B * Lq, E, generator=g)
    kfull = torch.randn(B * rows, E, generator=g)
    vfull = torch.randn(B * rows, 2 * E, generator=g)      # V lives in the right half of a wider buffer
    kpm = torch.zeros(B, Lk, dtype=torch.uint8)
    kpm[0, Lk // 2:] = 1
    dm = (torch.bernoulli(torch.full((B, H, Lq, Lk), 0.9), generator=g) / 0.9) if drop else None
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, kfull, vfull))
    k3 = kr.view(B, rows, E)[:, rows - Lk:]
    v3 = vr.view(B, rows, 2 * E)[:, rows - Lk:, E:]
    qh = qr.view(B, Lq, H, 32).transpose(1, 2) * 32 ** -0.5
    w = qh @ k3.reshape(B, Lk, H, 32).transpose(1, 2).transpose(-1, -2)
    w = w.masked_fill(kpm.bool()[:, None, None, :], float("-inf")).softmax(-1)
    if drop:
        w = w * dm
    o_ref = (w @ v3.reshape(B, Lk, H, 32).transpose(1, 2)).transpose(1, 2).reshape(B * Lq, E)
    do = torch.randn(B * Lq, E, generator=g)
    o_ref.backward(do)
    qd, kd, vd = (t.clone().to(DEV).requires_grad_(True) for t in (q, kfull, vfull))
    off = 1 if kv_off else 0
    o = SmallAttention.apply(qd, kd[off:], vd[off:, E:], B, H, Lq, Lk, kpm.to(DEV), None if dm is None else dm.to(DEV), rows)
    o.backward(do.to(DEV))
    close(o, o_ref, 1e-4, "out"); close(qd.grad, qr.grad, 1e-4, "dq"); close(kd.grad, kr.grad, 1e-4, "dk"); close(vd.grad, vr.grad, 1e-4, "dv")


def _rand_case(L, B, nq, TM, seed):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(L, B, nq, 2, generator=g)
    cxcy = torch.rand(L, B, nq, 2, generator=g) * 0.6 + 0.2
    wh = torch.rand(L, B, nq, 2, generator=g) * 0.3 + 0.05
    boxes = torch.cat([cxcy, wh], -1)
    tcount = torch.randint(0, min(TM, 5) + 1, (B,), generator=g).int()
    tcount[0] = 0 if B > 2 else tcount[0]
    tb = torch.cat([torch.rand(B, TM, 2, generator=g) * 0.6 + 0.2, torch.rand(B, TM, 2, generator=g) * 0.3 + 0.05], -1)
    tl = torch.zeros(B, TM, dtype=torch.int32)
    return logits, boxes, tb, tl, tcount


@pytest.mark.parametrize("L,B,nq", [(3, 8, 1), (3, 9, 10), (1, 5, 10), (2, 4, 3)])
def test_matcher_vs_scipy_and_criterion_vs_autograd(L, B, nq):
    from oracle import simvg_cpu as O
    from simvg_amd import hip_ops as ops
    TM = 16
    cfg = O.make_cfg("tiny", nq, 128)
    logits, boxes, tb, tl, tcount = _rand_case(L, B, nq, TM, 1000 + nq + B)
    targets = [{"labels": torch.zeros(int(tcount[b]), dtype=torch.long), "boxes": tb[b, :int(tcount[b])]} for b in range(B)]
    m = ops.match(logits.to(DEV), boxes.to(DEV), tb.to(DEV), tl.to(DEV), tcount.to(DEV)).cpu()
    for l in range(L):
        idx = O.hungarian(logits[l], boxes[l], targets, cfg)
        for b in range(B):
            exp = torch.full((nq,), -1, dtype=torch.int32)
            exp[idx[b][0]] = idx[b][1].int()
            assert torch.equal(m[l, b], exp), (l, b, m[l, b], exp)
    # criterion: value and gradients
    lg = logits.clone().requires_grad_(True)
    bx = boxes.clone().requires_grad_(True)
    losses = O.set_criterion(lg, bx, targets, cfg)
    coef = 2.0
    total = coef * sum(losses.values())
    total.backward()
    nb = torch.tensor([float(tcount.sum())])
    out, dl, db = ops.criterion(logits.to(DEV), boxes.to(DEV), m.to(DEV), tb.to(DEV), tl.to(DEV), nb.to(DEV), None, 0, coef)
    assert abs(float(out[0]) - float(total)) <= 1e-4 * max(1.0, abs(float(total)))
    close(dl, lg.grad, 1e-4, "dlogits")
    close(db, bx.grad, 1e-4, "dboxes")
    # per-layer terms: out[1+3l..] follows layer order l; the oracle keys: final = no suffix, aux i = _i
    for l in range(L):
        suf = "" if l == L - 1 else f"_{l}"
        for j, key in enumerate(["loss_class", "loss_bbox", "loss_giou"]):
            assert abs(float(out[1 + 3 * l + j]) - float(losses[key + suf])) <= 1e-4 * max(1.0, abs(float(losses[key + suf])))


def test_gemm_f32_grouped_matches_single_launches():
    """several independent problems (small-tile and 64x64-tile ones mixed, strided operands, bias / relu / addend /
    accumulate epilogues) in one launch == the same problems launched one by one, bit for bit"""
    from simvg_amd import hip_ops as ops
    g = torch.Generator().manual_seed(11)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)
    x, W, b, dy = r(64, 256), r(768, 256), r(768), r(64, 768)
    big_x, big_w = r(1280, 256), r(256, 256)
    add = r(64, 256)
    acc0 = r(256, 256)
    ones = torch.ones(64, device=DEV)

    def problems(outs):
        return [
            ops.gp(x, 256, 1, W, 1, 256, outs[0], 64, 512, 256, bias=b),                               # q|k projection
            ops.gp(x, 256, 1, W[512:], 1, 256, outs[0][:, 512:], 64, 256, 256, bias=b[512:], act=2),    # v, column view
            ops.gp(dy, 1, 768, x, 256, 1, outs[1], 768, 256, 64),                                       # wgrad
            ops.gp(ones, 0, 1, dy, 768, 1, outs[2], 1, 768, 64),                                        # bias gradient
            ops.gp(dy, 768, 1, W, 256, 1, outs[3], 64, 256, 768, addend=add, addend_rows=64),           # dgrad + residual
            ops.gp(big_x, 256, 1, big_w, 1, 256, outs[4], 1280, 256, 256),                              # 64x64-tile problem
            ops.gp(x, 1, 256, x, 256, 1, outs[5], 256, 256, 64, accumulate=True),                       # C += x^T x
        ]

    def outs():
        return [torch.zeros(64, 768, device=DEV), torch.zeros(768, 256, device=DEV), torch.zeros(1, 768, device=DEV),
                torch.zeros(64, 256, device=DEV), torch.zeros(1280, 256, device=DEV), acc0.clone()]

    one, grp = outs(), outs()
    for q in problems(one):
        ops.gemm_f32(q["A"], q["sam"], q["sak"], q["B"], q["sbk"], q["sbn"], q["C"], q["M"], q["N"], q["K"],
                     bias=q.get("bias"), addend=q.get("addend"), addend_rows=q.get("addend_rows", 0),
                     accumulate=q.get("accumulate", False), act=q.get("act", 0))
    ops.gemm_f32_group(problems(grp))
    for i, (a, c) in enumerate(zip(one, grp)):
        if i == 4:      # the 1280-row problem: in a group its K is split over two workgroups per tile (another summation order)
            close(c, a.cpu(), 1e-5, "1280 x 256 x 256")
        else:
            assert torch.equal(a, c)
    close(grp[1], dy.cpu().t() @ x.cpu(), 1e-5, "wgrad")
    close(grp[3], dy.cpu() @ W.cpu() + add.cpu(), 1e-5, "dgrad + addend")
    close(grp[5], acc0.cpu() + x.cpu().t() @ x.cpu(), 1e-5, "accumulate")
    # more than 12 problems: split into several launches
    many = [torch.zeros(64, 256, device=DEV) for _ in range(15)]
    ops.gemm_f32_group([ops.gp(x, 256, 1, big_w, 1, 256, o, 64, 256, 256) for o in many])
    for o in many:
        close(o, x.cpu() @ big_w.cpu().t(), 1e-5, "15 problems")


def test_gemm_f32_mid_size_problems_vectorised_tile_and_split_k():
    """The num_queries = 10 sizes (M = 640 query rows: forward Linears, dgrads with row-contiguous weights, weight gradients over
    the rows): 16-byte-load 64 x 64 tiles in all four operand orientations, K split over several workgroups per tile with the
    partial tiles added in a fixed order by a second launch -- against float64, with every epilogue option, repeatable bit for
    bit, and the split (workspace) and unsplit (no workspace) forms within fp32 summation-order distance of each other."""
    from simvg_amd import hip_ops as ops
    from simvg_amd import _lib
    import ctypes as C
    g = torch.Generator().manual_seed(21)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)
    R, E, Fh = 640, 256, 2048
    x, pos, h, dh, dy = r(R, E), r(R, E), r(R, Fh), r(R, Fh), r(R, E)
    W1, b1, W2, b2 = r(Fh, E) * E ** -0.5, r(Fh), r(E, Fh) * Fh ** -0.5, r(E)
    mult, gate, add, acc0 = (torch.rand(R, E, generator=g) > 0.1).float().to(DEV) / 0.9, r(R, Fh), r(R, E), r(Fh, E)
    mem = r(1280, E)

    def problems(o):
        return [
            ops.gp(h, Fh, 1, W2, 1, Fh, o[0], R, E, Fh, bias=b2, mult=mult, addend=add, addend_rows=R),      # KK, 40 tiles, K = 2048: split
            ops.gp(dy, E, 1, W2, Fh, 1, o[1], R, Fh, E, gate=gate),                                            # KN, 320 tiles: no split
            ops.gp(dh, Fh, 1, W1, E, 1, o[2], R, E, Fh, addend=add, addend_rows=R),                            # KN, K = 2048: split
            ops.gp(dh, 1, Fh, x, E, 1, o[3], Fh, E, R, accumulate=True),                                       # NN (weight gradient), split
            ops.gp(dy, 1, E, h, Fh, 1, o[4], E, Fh, R),                                                        # NN, 128 tiles
            ops.gp(x, E, 1, W1, 1, E, o[5], R, Fh, E, bias=b1, act=2, A2=pos),                                 # KK, (x + pos) W^T, relu
            ops.gp(W1, 1, E, dh, 1, Fh, o[6], E, R, Fh),                                                       # NK: A row-contiguous, B K-contiguous
            ops.gp(mem, E, 1, W1[:E], 1, E, o[7], 1280, E, E, B2=W1[E:2 * E]),                                 # KK with a second B operand
            ops.gp(torch.ones(R, device=DEV), 0, 1, dh, Fh, 1, o[8], 1, Fh, R),                                # bias gradient: small kernel beside them
        ]

    def outs():
        return [torch.zeros(R, E, device=DEV), torch.zeros(R, Fh, device=DEV), torch.zeros(R, E, device=DEV), acc0.clone(),
                torch.zeros(E, Fh, device=DEV), torch.zeros(R, Fh, device=DEV), torch.zeros(E, R, device=DEV),
                torch.zeros(1280, E, device=DEV), torch.zeros(1, Fh, device=DEV)]

    d = lambda t: t.double().cpu()
    ref = [
        (d(h) @ d(W2).t() + d(b2)) * d(mult) + d(add),
        torch.where(d(gate) > 0, d(dy) @ d(W2), torch.zeros((), dtype=torch.float64)),
        d(dh) @ d(W1) + d(add),
        d(acc0) + d(dh).t() @ d(x),
        d(dy).t() @ d(h),
        torch.relu((d(x) + d(pos)) @ d(W1).t() + d(b1)),
        d(W1).t() @ d(dh).t(),
        d(mem) @ (d(W1[:E]) + d(W1[E:2 * E])).t(),
        d(dh).sum(0, keepdim=True),
    ]
    a, b = outs(), outs()
    ops.gemm_f32_group(problems(a))
    ops.gemm_f32_group(problems(b))
    for i, (u, v, w) in enumerate(zip(a, b, ref)):
        assert torch.equal(u, v), f"problem {i} not repeatable"
        err = float((u.double().cpu() - w).abs().max() / w.abs().max())
        assert err <= 5e-6, (i, err)
    # without a workspace nothing is split: the same results up to the order of the fp32 partial sums
    c = outs()
    pr = problems(c)
    arr = (_lib.GemmF32Problem * len(pr))()
    for dsc, q in zip(arr, pr):
        dsc.A, dsc.sam, dsc.sak, dsc.B, dsc.sbk, dsc.sbn = q["A"].data_ptr(), q["sam"], q["sak"], q["B"].data_ptr(), q["sbk"], q["sbn"]
        dsc.C, dsc.ldc, dsc.M, dsc.N, dsc.K = q["C"].data_ptr(), q["C"].stride(0), q["M"], q["N"], q["K"]
        for k, f in (("bias", "bias"), ("addend", "addend"), ("A2", "A2"), ("B2", "B2"), ("mult", "mult"), ("gate", "gate")):
            setattr(dsc, f, q[k].data_ptr() if q.get(k) is not None else None)
        dsc.ld_addend = q["addend"].stride(0) if q.get("addend") is not None else 0
        dsc.addend_rows = q.get("addend_rows", 0)
        dsc.ld_mult = q["mult"].stride(0) if q.get("mult") is not None else 0
        dsc.ld_gate = q["gate"].stride(0) if q.get("gate") is not None else 0
        dsc.accumulate, dsc.act = int(q.get("accumulate", False)), q.get("act", 0)
    _lib.check(_lib.load().simvg_gemm_f32_grouped(C.byref(arr), len(pr), ops._stream()), "simvg_gemm_f32_grouped")
    for i, (u, w) in enumerate(zip(c, ref)):
        err = float((u.double().cpu() - w).abs().max() / w.abs().max())
        assert err <= 5e-6, ("unsplit", i, err)
    # the single-problem entry point takes the same tile for mid-size problems with enough tiles
    o = torch.zeros(R, Fh, device=DEV)
    ops.gemm_f32(x, E, 1, W1, 1, E, o, R, Fh, E, bias=b1, act=2)
    assert float((o.double().cpu() - torch.relu(d(x) @ d(W1).t() + d(b1))).abs().max()) <= 2e-5
    # ragged edges: M not a multiple of 64, N not a multiple of 64
    o2 = torch.zeros(600, 200, device=DEV)
    ops.gemm_f32_group([ops.gp(h[:600], Fh, 1, W2[:200], 1, Fh, o2, 600, 200, Fh, bias=b2[:200])])
    w2 = d(h[:600]) @ d(W2[:200]).t() + d(b2[:200])
    assert float((o2.double().cpu() - w2).abs().max() / w2.abs().max()) <= 2e-6


def _mha_ref(xq, xk, xv, W, b, B, H, Lq, Lk, kpm=None, dm=None):
    """nn.MultiheadAttention core in plain fp32 PyTorch: in_proj rows [q | k | v], heads of E/H, scale d^-1/2"""
    E = xq.shape[1]
    q = F.linear(xq, W[:E], b[:E]).view(B, Lq, H, E // H).transpose(1, 2) * (E // H) ** -0.5
    k = F.linear(xk, W[E:2 * E], b[E:2 * E]).view(B, Lk, H, E // H).transpose(1, 2)
    v = F.linear(xv, W[2 * E:], b[2 * E:]).view(B, Lk, H, E // H).transpose(1, 2)
    w = q @ k.transpose(-1, -2)
    if kpm is not None:
        w = w.masked_fill(kpm.bool()[:, None, None, :], float("-inf"))
    w = w.softmax(-1)
    if dm is not None:
        w = w * dm
    return (w @ v).transpose(1, 2).reshape(B * Lq, E)


def _ref_decoder_layer(tgt, qpos, P, B, H, nq, kind, xk=None, xv=None, mem=None, pos=None, kpm=None, post=None, masks=None):
    """post-norm DETR decoder layer in plain fp32 PyTorch; P = the 18 layer parameters; masks = None (eval) or the dropout
    multipliers (dm0, dm1 on the attention probabilities, m1 after the ReLU, m2 on the FFN output)"""
    dm0, dm1, m1, m2 = masks if masks is not None else (None, None, 1.0, 1.0)
    (Ws, bs, Wso, bso, g0, b0, Wc, bc, Wco, bco, g1, b1n, W1, b1, W2, b2, g2, b2n) = P
    E = tgt.shape[1]
    x_qk = tgt + qpos
    o = _mha_ref(x_qk, x_qk, tgt, Ws, bs, B, H, nq, nq, dm=dm0)
    t1 = F.layer_norm(tgt + F.linear(o, Wso, bso), (E,), g0, b0, 1e-5)
    xq = t1 + qpos
    if kind == "text":
        Lk = xk.shape[0] // B
        o2 = _mha_ref(xq, xk, xv, Wc, bc, B, H, nq, Lk, kpm=kpm, dm=dm1)
    else:
        Nv = mem.shape[0] // B
        patches = mem.view(B, Nv, E)[:, 1:]
        keys = (patches + (pos[None] if pos.dim() == 2 else pos)).reshape(-1, E)
        o2 = _mha_ref(xq, keys, patches.reshape(-1, E), Wc, bc, B, H, nq, Nv - 1, kpm=kpm, dm=dm1)
    t2 = F.layer_norm(t1 + F.linear(o2, Wco, bco), (E,), g1, b1n, 1e-5)
    t3 = F.layer_norm(t2 + F.linear(F.relu(F.linear(t2, W1, b1)) * m1, W2, b2) * m2, (E,), g2, b2n, 1e-5)
    hs = F.layer_norm(t3, (E,), post[0], post[1], 1e-5) if post is not None else None
    return t3, hs


@pytest.mark.parametrize("kind,nq,shared_pos", [("text", 1, True), ("text", 10, True), ("mem", 3, True), ("mem", 1, False)])
def test_decoder_layer_node_fwd_bwd(kind, nq, shared_pos):
    """The whole decoder layer as ONE autograd node (csrc/decoder.hip: three launches forward, four backward) against a
    plain PyTorch layer: both outputs and the gradient of every input and parameter.  fp32 memory for the "mem" flavour
    (the 16-bit-memory flavour: tests/test_decoder_gpu.py, and end to end against the reference model in test_model_gpu)."""
    from simvg_amd.models.heads.functions import DecoderLayerFn, LayerCfg
    B, H, E, Fd, T, HW = 3, 8, 256, 512, 20, 16
    Nv = HW + 1
    g = torch.Generator().manual_seed(17 + nq)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    tgt, qpos = r(B * nq, E), r(B * nq, E)
    P = [r(3 * E, E, sc=E ** -0.5), r(3 * E, sc=0.1), r(E, E, sc=E ** -0.5), r(E, sc=0.1), 1 + r(E, sc=0.1), r(E, sc=0.1),
         r(3 * E, E, sc=E ** -0.5), r(3 * E, sc=0.1), r(E, E, sc=E ** -0.5), r(E, sc=0.1), 1 + r(E, sc=0.1), r(E, sc=0.1),
         r(Fd, E, sc=E ** -0.5), r(Fd, sc=0.1), r(E, Fd, sc=Fd ** -0.5), r(E, sc=0.1), 1 + r(E, sc=0.1), r(E, sc=0.1)]
    post = [1 + r(E, sc=0.1), r(E, sc=0.1)]
    kpm = None
    if kind == "text":
        src = r(B * T, E)                          # the text rows: values; keys = rows + the 1-D position table
        pos = r(T, E)
        kpm = torch.zeros(B, T, dtype=torch.uint8)
        kpm[1, 12:] = 1
    else:
        src = r(B * Nv, E)                         # the image memory: row 0 of a sample is its CLS row, never a key
        pos = r(HW, E) if shared_pos else r(B, HW, E)
        if not shared_pos:
            kpm = torch.zeros(B, HW, dtype=torch.uint8)
            kpm[0, 10:] = 1
    d_t3, d_hs = r(B * nq, E), r(B * nq, E)

    def leaves(ts, dev):
        return [None if t is None else t.clone().to(dev).requires_grad_(True) for t in ts]

    ins_c = leaves([tgt, qpos, src] + P + post, "cpu")
    ref_kw = dict(xk=ins_c[2] + pos.repeat(B, 1), xv=ins_c[2]) if kind == "text" else dict(mem=ins_c[2], pos=pos)
    t3_ref, hs_ref = _ref_decoder_layer(ins_c[0], ins_c[1], ins_c[3:21], B, H, nq, kind, kpm=kpm, post=ins_c[21:23], **ref_kw)
    torch.autograd.backward([t3_ref, hs_ref], [d_t3, d_hs])
    ins_d = leaves([tgt, qpos, src] + P + post, DEV)
    cfg = LayerCfg(B, H, nq, kind, T if kind == "text" else HW, kpm=None if kpm is None else kpm.to(DEV),
                   pos=pos.to(DEV), Nv=Nv if kind == "mem" else 0, training=False)
    t3, hs = DecoderLayerFn.apply(*ins_d, cfg)
    torch.autograd.backward([t3, hs], [d_t3.to(DEV), d_hs.to(DEV)])
    close(t3, t3_ref, 2e-5, "layer output"); close(hs, hs_ref, 2e-5, "post-normed output")
    names = ["tgt", "qpos", "src", "self.in_w", "self.in_b", "self.out_w", "self.out_b", "norm0.w", "norm0.b",
             "cross.in_w", "cross.in_b", "cross.out_w", "cross.out_b", "norm1.w", "norm1.b", "ffn.w1", "ffn.b1", "ffn.w2",
             "ffn.b2", "norm2.w", "norm2.b", "post.w", "post.b"]
    for n, d, c in zip(names, ins_d, ins_c):
        if n == "cross.in_b":
            # the key third of the in-projection bias moves every score of a row alike: its gradient is exactly zero (the fused
            # kernel writes zeros; autograd leaves rounding noise)
            assert float(d.grad[E:2 * E].abs().max()) == 0.0
            close(torch.cat([d.grad[:E], d.grad[2 * E:]]), torch.cat([c.grad[:E], c.grad[2 * E:]]), 1e-4, "grad " + n)
            continue
        close(d.grad, c.grad, 1e-4, "grad " + n)
    if kind == "mem":
        assert float(ins_d[2].grad.view(B, Nv, E)[:, 0].abs().max()) == 0.0        # CLS rows are not keys: no gradient


def test_gemm_f32_operand_sums_multiplier_and_gate():
    """the optional pieces of a grouped problem: A + A2 / B + B2 formed on the operand load (both tile shapes, both operand
    orientations), `mult` after the activation, `gate` on a saved output, addend added after them"""
    from simvg_amd import hip_ops as ops
    g = torch.Generator().manual_seed(23)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)
    for M in (64, 1280):                                    # 16x16-tile and 64x64-tile kernels
        x, x2, W, b = r(M, 256), r(M, 256), r(512, 256), r(512)
        mult = (torch.rand(M, 512, generator=g) > 0.3).float().to(DEV) / 0.7
        saved, res, dy = r(M, 512), r(M, 512), r(M, 512)
        y0, y1, y2 = (torch.zeros(M, 512, device=DEV) for _ in range(3))
        dW = torch.zeros(512, 256, device=DEV)
        ops.gemm_f32_group([
            ops.gp(x, 256, 1, W, 1, 256, y0, M, 512, 256, bias=b, A2=x2),                                          # (x + x2) W^T + b
            ops.gp(x, 256, 1, W, 1, 256, y1, M, 512, 256, bias=b, act=2, mult=mult, addend=res, addend_rows=M),    # relu(.)*m + res
            ops.gp(x, 256, 1, W, 1, 256, y2, M, 512, 256, gate=saved, mult=mult),                                  # gated
            ops.gp(dy, 1, 512, x, 256, 1, dW, 512, 256, M, B2=x2)])                                                # dy^T (x + x2)
        xc, x2c, Wc = x.cpu().double(), x2.cpu().double(), W.cpu().double()
        close(y0, ((xc + x2c) @ Wc.t() + b.cpu().double()).float(), 2e-5, f"A2 M={M}")
        close(y1, (F.relu(xc @ Wc.t() + b.cpu().double()) * mult.cpu().double() + res.cpu().double()).float(), 2e-5, f"mult M={M}")
        close(y2, ((xc @ Wc.t()) * mult.cpu().double() * (saved.cpu() > 0).double()).float(), 2e-5, f"gate M={M}")
        close(dW, (dy.cpu().double().t() @ (xc + x2c)).float(), 2e-5 * (M / 64) ** 0.5, f"B2 M={M}")


@pytest.mark.parametrize("kind,nq", [("text", 3), ("mem", 2)])
def test_decoder_layer_node_with_dropout_multipliers(kind, nq):
    """training branch of the layer node: the four dropout sites take their multipliers from `mask_fn` (attention
    probabilities, after the ReLU and on the FFN output through the GEMM epilogues) == the plain PyTorch layer with the
    same multipliers, outputs and every gradient"""
    from simvg_amd.models.heads.functions import DecoderLayerFn, LayerCfg
    B, H, E, Fd, T, HW = 3, 8, 256, 512, 20, 16
    Nv = HW + 1
    g = torch.Generator().manual_seed(41 + nq)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    tgt, qpos = r(B * nq, E), r(B * nq, E)
    P = [r(3 * E, E, sc=E ** -0.5), r(3 * E, sc=0.1), r(E, E, sc=E ** -0.5), r(E, sc=0.1), 1 + r(E, sc=0.1), r(E, sc=0.1),
         r(3 * E, E, sc=E ** -0.5), r(3 * E, sc=0.1), r(E, E, sc=E ** -0.5), r(E, sc=0.1), 1 + r(E, sc=0.1), r(E, sc=0.1),
         r(Fd, E, sc=E ** -0.5), r(Fd, sc=0.1), r(E, Fd, sc=Fd ** -0.5), r(E, sc=0.1), 1 + r(E, sc=0.1), r(E, sc=0.1)]
    post = [1 + r(E, sc=0.1), r(E, sc=0.1)]
    Lk = T if kind == "text" else HW
    if kind == "text":
        src, pos = r(B * T, E), r(T, E)
    else:
        src, pos = r(B * Nv, E), r(HW, E)
    keep = lambda *s: (torch.rand(*s, generator=g) > 0.1).float() / 0.9
    masks = [keep(B, H, nq, nq), keep(B, H, nq, Lk), keep(B * nq, Fd), keep(B * nq, E)]
    d_t3, d_hs = r(B * nq, E), r(B * nq, E)

    def leaves(ts, dev):
        return [None if t is None else t.clone().to(dev).requires_grad_(True) for t in ts]

    ins_c = leaves([tgt, qpos, src] + P + post, "cpu")
    ref_kw = dict(xk=ins_c[2] + pos.repeat(B, 1), xv=ins_c[2]) if kind == "text" else dict(mem=ins_c[2], pos=pos)
    t3_ref, hs_ref = _ref_decoder_layer(ins_c[0], ins_c[1], ins_c[3:21], B, H, nq, kind, post=ins_c[21:23], masks=masks, **ref_kw)
    torch.autograd.backward([t3_ref, hs_ref], [d_t3, d_hs])
    ins_d = leaves([tgt, qpos, src] + P + post, DEV)
    order = iter([m.to(DEV) for m in masks])            # the node asks in the order: self-attn, cross-attn, ffn 1, ffn 2

    def mask_fn(shape, dev, p=None):
        m = next(order)
        assert tuple(m.shape) == tuple(shape), (m.shape, shape)
        return m

    cfg = LayerCfg(B, H, nq, kind, Lk, pos=pos.to(DEV), Nv=Nv if kind == "mem" else 0,
                   p_attn=0.1, p_ffn=0.1, training=True, mask_fn=mask_fn)
    t3, hs = DecoderLayerFn.apply(*ins_d, cfg)
    torch.autograd.backward([t3, hs], [d_t3.to(DEV), d_hs.to(DEV)])
    close(t3, t3_ref, 2e-5, "layer output"); close(hs, hs_ref, 2e-5, "post-normed output")
    for i, (d, c) in enumerate(zip(ins_d, ins_c)):
        if i == 10:          # cross-attention in-projection bias: the key third's gradient is exactly zero in the fused kernel
            close(torch.cat([d.grad[:E], d.grad[2 * E:]]), torch.cat([c.grad[:E], c.grad[2 * E:]]), 1e-4, f"grad of input {i}")
            continue
        close(d.grad, c.grad, 1e-4, f"grad of input {i}")


@pytest.mark.parametrize("B,nq,ncol,rescale", [(5, 1, 2, False), (4, 10, 2, True), (3, 7, 4, True)])
def test_postprocess_kernel_vs_torch(B, nq, ncol, rescale):
    """`simvg_postprocess` against the reference's formulation in plain PyTorch: head.inference (softmax, drop the last
    column, cxcywh -> xyxy * (w,h,w,h)), detector_postprocess (clip, nonempty), best kept query per image, / scale_factor"""
    from simvg_amd import hip_ops as ops
    g = torch.Generator().manual_seed(B * 31 + nq)
    logits = torch.randn(B, nq, ncol, generator=g) * 2
    boxes = torch.rand(B, nq, 4, generator=g)
    boxes[0, 0] = torch.tensor([0.5, 0.5, 0.0, 0.3])            # empty box -> dropped
    boxes[1, :, 2:] *= 0.0                                       # no query of image 1 survives -> query 0 is reported
    boxes[2, 0] = torch.tensor([0.95, 0.9, 0.4, 0.5])            # sticks out of the image -> clipped
    hw = torch.tensor([[480, 640], [640, 640], [333, 500], [640, 427], [512, 512]])[:B]
    lim = torch.stack([hw[:, 1], hw[:, 0], hw[:, 1], hw[:, 0]], 1).float()
    sf = (torch.rand(B, 4, generator=g) + 0.5) if rescale else None
    # torch formulation
    prob = F.softmax(logits, -1)[:, :, :-1]
    scores_ref, labels_ref = prob.max(-1)
    cx, cy, w, h = boxes.unbind(-1)
    xyxy = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1) * lim[:, None, :]
    xyxy = torch.minimum(xyxy.clamp(min=0), lim[:, None, :])
    keep_ref = ((xyxy[..., 2] - xyxy[..., 0]) > 0) & ((xyxy[..., 3] - xyxy[..., 1]) > 0)
    best = torch.where(keep_ref, scores_ref, torch.full_like(scores_ref, -1.0)).argmax(1)
    box_ref = xyxy[torch.arange(B), best]
    if sf is not None:
        box_ref, xyxy = box_ref / sf, xyxy / sf[:, None, :]
    scores, labels, out_xyxy, keep, box, best_label = ops.postprocess(logits.to(DEV), boxes.to(DEV), lim.to(DEV),
                                                                     None if sf is None else sf.to(DEV))
    assert torch.equal(keep.cpu(), keep_ref) and not keep_ref[0, 0] and not keep_ref[1].any()
    assert torch.equal(labels.cpu(), labels_ref)
    assert torch.equal(out_xyxy.cpu(), xyxy)                     # exact fp32 arithmetic, same operation order
    assert torch.equal(box.cpu(), box_ref)
    assert torch.equal(best_label.cpu(), labels_ref[torch.arange(B), best])
    close(scores, scores_ref, 1e-6, "scores")


def test_pack_targets_matches_the_reference_formula():
    """`simvg_pack_targets` (one copy + one launch) against prepare_soft_targets' arithmetic (tgqs_kd_detr_head.py:215-234): boxes
    in HBM (views of one batch tensor, as the loaders hand them over), boxes on the host, multi-target lists with dropped
    (category_id == -1) entries and images without a target -- bit-identical rows, zeros elsewhere, counts per image."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_tools_gpu import _tiny_model
    _, model = _tiny_model(3)
    head = model.head
    TM = head.max_targets
    g = torch.Generator().manual_seed(5)
    B = 6
    xy = torch.rand(B, 3, 2, generator=g) * 300
    wh = 20 + torch.rand(B, 3, 2, generator=g) * 200
    allb = torch.cat([xy, xy + wh], -1)                                   # [B, 3, 4] pixel xyxy
    shapes = [(480, 640), (640, 480), (333, 517), (640, 640), (512, 400), (200, 300)]
    metas, gts, ref = [], [], torch.zeros(B, TM, 4)
    counts = []
    for b in range(B):
        h, w = shapes[b]
        if b % 3 == 0:            # single box given as a 1-D tensor (RefCOCO): a view of the batch tensor in HBM
            gts.append(allb.to(DEV)[b, 0])
            metas.append(dict(img_shape=(h, w, 3)))
            kept = [allb[b, 0]]
        elif b % 3 == 1:          # GRefCOCO list on the host, second entry is a no-target placeholder
            gts.append(allb[b].clone())
            metas.append(dict(img_shape=(h, w, 3), target=[dict(category_id=1), dict(category_id=-1), dict(category_id=3)]))
            kept = [allb[b, 0], allb[b, 2]]
        else:                     # GRefCOCO list in HBM, nothing kept
            gts.append(allb[b].to(DEV))
            metas.append(dict(img_shape=(h, w, 3), target=[dict(category_id=-1)] * 3))
            kept = []
        counts.append(len(kept))
        for j, bx in enumerate(kept):
            t = bx / torch.tensor([w, h, w, h], dtype=torch.float32)
            ref[b, j] = torch.stack([(t[0] + t[2]) / 2, (t[1] + t[3]) / 2, t[2] - t[0], t[3] - t[1]])
    boxes, labels, count, host_counts = head._pack_targets(gts, metas, torch.device(DEV), return_counts=True)
    torch.cuda.synchronize()
    assert host_counts == counts and count.cpu().tolist() == counts
    assert labels.shape == (B, TM) and int(labels.abs().sum()) == 0
    assert torch.equal(boxes.cpu(), ref)
    # the staging ring survives more calls than it has slots
    for _ in range(20):
        b2 = head._pack_targets(gts, metas, torch.device(DEV))[0]
    assert torch.equal(b2.cpu(), ref)
