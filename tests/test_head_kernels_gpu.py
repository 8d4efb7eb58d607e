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
    for a, c in zip(one, grp):
        assert torch.equal(a, c)
    close(grp[1], dy.cpu().t() @ x.cpu(), 1e-5, "wgrad")
    close(grp[3], dy.cpu() @ W.cpu() + add.cpu(), 1e-5, "dgrad + addend")
    close(grp[5], acc0.cpu() + x.cpu().t() @ x.cpu(), 1e-5, "accumulate")
    # more than 12 problems: split into several launches
    many = [torch.zeros(64, 256, device=DEV) for _ in range(15)]
    ops.gemm_f32_group([ops.gp(x, 256, 1, big_w, 1, 256, o, 64, 256, 256) for o in many])
    for o in many:
        close(o, x.cpu() @ big_w.cpu().t(), 1e-5, "15 problems")


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


@pytest.mark.parametrize("B,L,drop", [(4, 1, False), (3, 10, True)])
def test_self_attn_block_fwd_bwd(B, L, drop):
    from simvg_amd.models.heads.functions import SelfAttnBlock
    H, E = 8, 256
    g = torch.Generator().manual_seed(7 * B + L)
    x_qk, x_v = torch.randn(B * L, E, generator=g), torch.randn(B * L, E, generator=g)
    W, b = torch.randn(3 * E, E, generator=g) * E ** -0.5, torch.randn(3 * E, generator=g) * 0.1
    dm = (torch.bernoulli(torch.full((B, H, L, L), 0.9), generator=g) / 0.9) if drop else None
    dout = torch.randn(B * L, E, generator=g)
    ref_in = [t.clone().requires_grad_(True) for t in (x_qk, x_v, W, b)]
    _mha_ref(ref_in[0], ref_in[0], ref_in[1], ref_in[2], ref_in[3], B, H, L, L, dm=dm).backward(dout)
    dev_in = [t.clone().to(DEV).requires_grad_(True) for t in (x_qk, x_v, W, b)]
    out = SelfAttnBlock.apply(*dev_in, B, H, L, None if dm is None else dm.to(DEV))
    out.backward(dout.to(DEV))
    close(out, _mha_ref(x_qk, x_qk, x_v, W, b, B, H, L, L, dm=dm), 1e-5, "out")
    for name, d, r in zip(("dx_qk", "dx_v", "dW", "db"), dev_in, ref_in):
        close(d.grad, r.grad, 2e-5, name)


def test_cross_attn_block_fwd_bwd():
    from simvg_amd.models.heads.functions import CrossAttnBlock
    B, H, E, Lq, Lk = 3, 8, 256, 10, 20
    g = torch.Generator().manual_seed(3)
    xq, xk, xv = (torch.randn(B * n, E, generator=g) for n in (Lq, Lk, Lk))
    W, b = torch.randn(3 * E, E, generator=g) * E ** -0.5, torch.randn(3 * E, generator=g) * 0.1
    kpm = torch.zeros(B, Lk, dtype=torch.uint8)
    kpm[1, 12:] = 1
    dm = torch.bernoulli(torch.full((B, H, Lq, Lk), 0.9), generator=g) / 0.9
    dout = torch.randn(B * Lq, E, generator=g)
    ref_in = [t.clone().requires_grad_(True) for t in (xq, xk, xv, W, b)]
    _mha_ref(*ref_in, B, H, Lq, Lk, kpm=kpm, dm=dm).backward(dout)
    dev_in = [t.clone().to(DEV).requires_grad_(True) for t in (xq, xk, xv, W, b)]
    out = CrossAttnBlock.apply(*dev_in, B, H, Lq, Lk, kpm.to(DEV), dm.to(DEV))
    out.backward(dout.to(DEV))
    close(out, _mha_ref(xq, xk, xv, W, b, B, H, Lq, Lk, kpm=kpm, dm=dm), 1e-5, "out")
    for name, d, r in zip(("dxq", "dxk", "dxv", "dW", "db"), dev_in, ref_in):
        close(d.grad, r.grad, 2e-5, name)


@pytest.mark.parametrize("shared_pos", [True, False])
def test_mem_cross_attn_block_fp32_fwd_bwd(shared_pos):
    """the exact (fp32-memory) flavour against plain PyTorch: key = memory + key_pos on the patch rows, value = memory,
    the CLS row of every sample is carried but never attended to"""
    from simvg_amd.models.heads.functions import MemCrossAttnBlock
    B, H, E, Lq, HW = 2, 8, 256, 3, 16
    Nv = HW + 1
    g = torch.Generator().manual_seed(9)
    xq, mem = torch.randn(B * Lq, E, generator=g), torch.randn(B * Nv, E, generator=g)
    pos = torch.randn(HW, E, generator=g) if shared_pos else torch.randn(B, HW, E, generator=g)
    W, b = torch.randn(3 * E, E, generator=g) * E ** -0.5, torch.randn(3 * E, generator=g) * 0.1
    kpm = None
    if not shared_pos:
        kpm = torch.zeros(B, HW, dtype=torch.uint8)
        kpm[0, 10:] = 1
    dout = torch.randn(B * Lq, E, generator=g)

    def ref(xq_, mem_, W_, b_):
        patches = mem_.view(B, Nv, E)[:, 1:]
        xk = (patches + (pos[None] if shared_pos else pos)).reshape(B * HW, E)
        return _mha_ref(xq_, xk, patches.reshape(B * HW, E), W_, b_, B, H, Lq, HW, kpm=kpm)

    ref_in = [t.clone().requires_grad_(True) for t in (xq, mem, W, b)]
    ref(*ref_in).backward(dout)
    dev_in = [t.clone().to(DEV).requires_grad_(True) for t in (xq, mem, W, b)]
    out = MemCrossAttnBlock.apply(dev_in[0], dev_in[1], dev_in[2], dev_in[3], pos.to(DEV), None, None, B, H, Lq, Nv,
                                  None if kpm is None else kpm.to(DEV), None)
    out.backward(dout.to(DEV))
    close(out, ref(xq, mem, W, b), 1e-5, "out")
    for name, d, r in zip(("dxq", "dmem", "dW", "db"), dev_in, ref_in):
        close(d.grad, r.grad, 2e-5, name)
    assert float(dev_in[1].grad.view(B, Nv, E)[:, 0].abs().max()) == 0.0      # CLS rows get no gradient
