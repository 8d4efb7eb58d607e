"""GPU (MI355X) op-level parity: every HIP kernel against fp32/fp64 torch math on the CPU, on the same
inputs, which are exactly representable in both 16-bit formats the library can be built for (bf16's 8 significand
bits are a subset of fp16's 11).  Tolerances: kernels accumulate in fp32 and store 16-bit values (fp16 half-ulp 2^-12,
bf16 2^-9 relative), so 16-bit outputs are compared at 2e-3 (fp16 build) / 1e-2 (bf16 build) of the reference scale
and fp32 outputs at 1e-3 (inputs to the MFMA are exact, products are exact in fp32)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from simvg_amd import hip_ops
    return hip_ops


def LPD():
    """torch dtype of the library's 16-bit format"""
    return _ops().LP()


def LPTOL(extra=1.0):
    """relative tolerance of a value stored in the 16-bit format (x `extra` for outputs of longer rounding chains)"""
    return (2e-3 if LPD() == torch.float16 else 1e-2) * extra


def bf(t):
    return t.to(LPD())


def rnd_bf16(*shape, scale=1.0, gen=None):
    """fp32 CPU tensor whose values are exactly representable in bf16."""
    return (torch.randn(*shape, generator=gen) * scale).to(torch.bfloat16).float()


def assert_close(got, ref, tol, what=""):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite values"
    scale = max(float(ref.abs().max()), 1e-6)
    err = float((got - ref).abs().max())
    assert err <= tol * scale, f"{what}: max err {err:.4g} vs scale {scale:.4g} (tol {tol})"


# ------------------------------------------------------------------------------------------
# hardware semantics the kernels rely on
# ------------------------------------------------------------------------------------------
def test_probe_mfma_layout():
    import ctypes as C
    from simvg_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    a = rnd_bf16(16, 32, gen=g)
    b = rnd_bf16(32, 16, gen=g)          # asymmetric on purpose
    out = torch.zeros(64, 4, device=DEV)
    ad, bd = bf(a).to(DEV), bf(b).to(DEV)
    _lib.check(lib.simvg_probe_mfma(C.c_void_p(ad.data_ptr()), C.c_void_p(bd.data_ptr()), C.c_void_p(out.data_ptr()),
                                    None), "probe_mfma")
    torch.cuda.synchronize()
    ref = a @ b
    got = torch.zeros(16, 16)
    o = out.cpu()
    for lane in range(64):
        for r in range(4):
            got[(lane >> 4) * 4 + r, lane & 15] = o[lane, r]   # C/D: col = lane&15, row = (lane>>4)*4 + r
    assert_close(got, ref, 1e-5, "mfma 16x16x32 C/D layout")


def test_probe_tr16_semantics():
    import ctypes as C
    from simvg_amd import _lib
    lib = _lib.load()
    stride = 144
    addr = torch.zeros(64, dtype=torch.int32)
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        addr[lane] = g * 1024 + (i >> 2) * stride + (i & 3) * 8
    out = torch.zeros(64, 4, dtype=torch.int16, device=DEV)
    ad = addr.to(DEV)
    _lib.check(lib.simvg_probe_tr16(C.c_void_p(ad.data_ptr()), C.c_void_p(out.data_ptr()), None), "probe_tr16")
    torch.cuda.synchronize()
    o = out.cpu()
    exp = torch.zeros(64, 4, dtype=torch.int16)
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        for r in range(4):
            exp[lane, r] = (g * 1024 + r * stride + i * 2) // 2   # lane i <- column i, rows 0..3
    if not torch.equal(o, exp):
        print("observed tr16 table (lane: 4 element indices):")
        for lane in range(64):
            print(lane, o[lane].tolist(), "expected", exp[lane].tolist())
    assert torch.equal(o, exp)


def test_probe_glds_lane_linear():
    import ctypes as C
    from simvg_amd import _lib
    lib = _lib.load()
    src = torch.arange(512, dtype=torch.int16)
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(1)).to(torch.int32)
    out = torch.zeros(512, dtype=torch.int16, device=DEV)
    sd, pd = src.to(DEV), perm.to(DEV)
    _lib.check(lib.simvg_probe_glds(C.c_void_p(sd.data_ptr()), C.c_void_p(pd.data_ptr()), C.c_void_p(out.data_ptr()),
                                    None), "probe_glds")
    torch.cuda.synchronize()
    exp = torch.cat([src[p * 8:(p + 1) * 8] for p in perm.tolist()])   # LDS image = lane-linear
    assert torch.equal(out.cpu(), exp)


# ------------------------------------------------------------------------------------------
# GEMMs
# ------------------------------------------------------------------------------------------
# (which kernel a shape / epilogue selects: `simvg_gemm_nt_plan`, pinned as data in tests/test_abi.py)
@pytest.mark.parametrize("M,N,K,split", [(300, 192, 128, 200), (1684, 768, 768, 1604), (257, 64, 64, 0),      # 64x64 latency kernel
                                         (640, 768, 3072, 512), (2230, 768, 256, 2000), (2048, 768, 3072, 0), (300, 7040, 64, 200),
                                         (1684, 3072, 768, 1604), (12000, 320, 64, 11000),                     # 256x128 / k 32 kernel
                                         (300, 12800, 64, 200),                                                # 128x128 kernel
                                         (2370, 2304, 768, 2100), (2112, 3072, 128, 0),                        # 160x256 kernel
                                         (2306, 4096, 64, 2100),      # ... 16 column tiles
                                         (7300, 768, 192, 6000), (9900, 1024, 128, 0),
                                         (7000, 3072, 128, 6500),     # 336 tiles of 256 rows: persistent workgroups take two (plain / bias), the
                                                                      # one-tile 256x256 kernel (other epilogues); walked in column groups, ragged row groups
                                         (10300, 1024, 256, 9800),    # one round of 224-row tiles: hand-managed 2 x 8 waves (round 6: plain / bias /
                                                                      # residual), compiler-scheduled (round 4: activations)
                                         # one round of 320-row tiles (round 6, gemm_nt_kernel_tall5_*: the plain / bias / residual modes):
                                         # BASELINE's N = 768 launch geometry (85 x 3 tiles); one round of 256-row hand-managed tiles for the
                                         # fp32 epilogue (tall4) beside the one-tile 256x256 kernel; one row group on 224-row tiles
                                         (26944, 768, 128, 25664), (20011, 768, 192, 18003), (16000, 768, 64, 0)])
@pytest.mark.parametrize("mode", ["plain", "bias", "bias_gelu_aux", "residual_scale_f32", "relu_f32"])
def test_gemm_nt(M, N, K, split, mode):
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    a = rnd_bf16(M, K, gen=g)
    ng = 2 if split else 1
    w = rnd_bf16(ng, N, K, scale=K ** -0.5, gen=g)
    bias = torch.randn(ng, N, generator=g)
    sp = split if split else M

    def ref_lin(with_bias):
        y = torch.cat([a[:sp] @ w[0].t(), a[sp:] @ w[-1].t()], 0)
        if with_bias:
            y = y + torch.cat([bias[0].expand(sp, N), bias[-1].expand(M - sp, N)], 0)
        return y

    ad, wd, bd = bf(a).to(DEV), bf(w).to(DEV), bias.to(DEV)
    if mode == "plain":
        out = ops.gemm_nt(ad, wd, split=split)
        assert_close(out, ref_lin(False), LPTOL(), "gemm plain")
    elif mode == "bias":         # with "plain": the epilogue of the persistent 256x256 kernel (bias = accumulator start value)
        out = ops.gemm_nt(ad, wd, bias=bd, split=split)
        assert_close(out, ref_lin(True), LPTOL(), "gemm bias")
    elif mode == "bias_gelu_aux":
        aux = torch.empty(M, N, device=DEV, dtype=LPD())
        out = ops.gemm_nt(ad, wd, bias=bd, split=split, act=1, aux_preact=aux)
        u = ref_lin(True)
        assert_close(aux, u, LPTOL(), "gemm aux preact")
        assert_close(out, F.gelu(u), LPTOL(), "gemm gelu")
    elif mode == "residual_scale_f32":
        rps = (50, 21) if split else (M, 1)
        nsamp = max(math.ceil(sp / rps[0]), math.ceil((M - sp) / rps[1]) if split else 1)
        scale = torch.rand(nsamp, generator=g) + 0.5
        res = torch.randn(M, N, generator=g)
        rows = torch.arange(M)
        samp = torch.where(rows < sp, rows // rps[0], (rows - sp) // rps[1])
        out = ops.gemm_nt(ad, wd, bias=bd, split=split, residual=res.to(DEV), row_scale=scale.to(DEV),
                          rows_per_sample=rps, out_dtype=torch.float32)
        assert_close(out, res + scale[samp][:, None] * ref_lin(True), 1e-3, "gemm residual")
    else:
        out = ops.gemm_nt(ad, wd, bias=bd, split=split, act=2, out_dtype=torch.float32)
        assert_close(out, F.relu(ref_lin(True)), 1e-3, "gemm relu f32")


@pytest.mark.parametrize("M,N,K,split,part", [(26944, 768, 256, 25664, 6000),    # one round of 320-row tiles vs the 160-row kernel
                                              (13472, 1024, 256, 12832, 3000),   # the 2 x 8-wave 224-row kernel vs the 256-row / 160-row ones
                                              (26944, 2304, 128, 25664, 5000)])  # persistent 256-row kernel at both sizes
def test_gemm_nt_rows_do_not_depend_on_the_kernel_their_batch_selects(M, N, K, split, part):
    """A row of a Linear is the same dot products in the same k order whichever tile extent / kernel the problem's row count selects
    (round 6: the one-round 320-row and 224-row kernels start their accumulators at zero and add the bias in the epilogue like the
    other one-tile kernels; every fp32 epilogue is fmaf(row_scale, acc + bias, residual)): the first `part` vision rows and all text
    rows of the big problem == the same rows computed as a small problem, bit for bit.  What forward_test's batch independence
    (tests/test_properties_gpu.py) rests on."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K + 7)
    a = bf(rnd_bf16(M, K, gen=g)).to(DEV)
    w = bf(rnd_bf16(2, N, K, scale=K ** -0.5, gen=g)).to(DEV)
    bias = torch.randn(2, N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV)
    nt = M - split
    rows = torch.cat([torch.arange(part), torch.arange(split, M)]).to(DEV)
    a_s, res_s = a[rows].contiguous(), res[rows].contiguous()
    for kw, kw_s in ((dict(residual=res, out_dtype=torch.float32), dict(residual=res_s, out_dtype=torch.float32)),
                     (dict(), dict())):
        if N > 1024 and kw:
            continue                      # (the wide launches of the path have 16-bit outputs)
        big = ops.gemm_nt(a, w, bias=bias, split=split, **kw)
        small = ops.gemm_nt(a_s, w, bias=bias, split=part, **kw_s)
        assert small.shape[0] == part + nt
        assert torch.equal(big[rows], small), (M, N, K, "fp32 + residual" if kw else "16-bit",
                                               float((big[rows].float() - small.float()).abs().max()))


@pytest.mark.parametrize("M,N,K,split", [(300, 192, 128, 200),        # latency kernel (64x64 tiles)
                                         (640, 768, 3072, 512),       # 256x128 / k 32 kernel
                                         (300, 7040, 64, 200),        # 128x128 kernel
                                         (2370, 2304, 768, 2100),     # 256x256 kernel (the persistent form is refused)
                                         (7300, 768, 192, 6000),                               # 160x256 kernel
                                         (7000, 2304, 128, 6500),                              # persistent 256x256 kernel, split variant (round 6)
                                         (26944, 768, 128, 25664)])                            # one round of 320-row tiles
@pytest.mark.parametrize("mode", ["bias_lp", "bias_residual_f32"])
def test_gemm_nt_split_carries_fp32_weights(M, N, K, split, mode):
    """simvg_gemm_nt_split: the weight as a hi + lo pair of 16-bit numbers (rows [lo * 2^11 | hi], A walked twice, accumulators
    rescaled between the halves).  Against fp32-weight math on the CPU the fp32 output agrees to 2e-5 of its scale (a single
    16-bit weight: 3e-4 in the fp16 build on the same data), and the 16-bit output to the format's own rounding."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K + 1)
    a = rnd_bf16(M, K, gen=g)
    ng = 2 if split else 1
    w = torch.randn(ng, N, K, generator=g) * K ** -0.5            # full fp32 weights: NOT representable in 16 bits
    bias = torch.randn(ng, N, generator=g)
    sp = split if split else M
    ref = torch.cat([a[:sp].double() @ w[0].double().t() + bias[0].double(), a[sp:].double() @ w[-1].double().t() + bias[-1].double()], 0)
    ad, bd = bf(a).to(DEV), bias.to(DEV)
    w2 = ops.split_weight(w.to(DEV))
    assert w2.shape == (ng, N, 2 * K) and w2.dtype == LPD()
    if mode == "bias_lp":
        out = ops.gemm_nt_split(ad, w2, bias=bd, split=split)
        assert_close(out, ref.float(), LPTOL(), "split gemm, 16-bit output")
    else:
        res = torch.randn(M, N, generator=g)
        out = ops.gemm_nt_split(ad, w2, bias=bd, split=split, residual=res.to(DEV), out_dtype=torch.float32)
        one = ops.gemm_nt(ad, bf(w).to(DEV), bias=bd, split=split, residual=res.to(DEV), out_dtype=torch.float32)
        scale = float(ref.abs().max())
        e2 = float((out.double().cpu() - (ref + res.double())).abs().max()) / scale
        e1 = float((one.double().cpu() - (ref + res.double())).abs().max()) / scale
        print(f"[split gemm {M}x{N}x{K}] error of the fp32 output vs fp32-weight math: split {e2:.2e}, single 16-bit weight {e1:.2e}")
        assert e2 <= (2e-5 if LPD() == torch.float16 else 2e-4) and e2 < e1 / 4


@pytest.mark.parametrize("M,N,K,split", [(300, 192, 128, 200), (1684, 768, 768, 1604), (5000, 128, 256, 0),
                                         (1684, 3072, 768, 1604), (1684, 768, 3072, 1604), (3000, 2304, 768, 2500),
                                         # M >= 4096 and an encoder shape: the XCD-partitioned kernel (csrc/wgrad.hip) --
                                         # ragged row counts, a row-group boundary inside a block's share, no text rows
                                         (4133, 3072, 768, 3850), (6011, 2304, 768, 5614), (4500, 768, 3072, 0),
                                         (8421, 768, 3072, 8020), (4096, 3072, 768, 4090),
                                         # ViT-L encoder shapes: the generic 256 x 128 kernel (gemm.hip), row chunks + fp32 atomics
                                         (4700, 4096, 1024, 4400), (4517, 1024, 4096, 4100), (4500, 3072, 1024, 0)])
def test_gemm_tn_and_colsum(M, N, K, split):
    ops = _ops()
    g = torch.Generator().manual_seed(7 * M + N)
    dy = rnd_bf16(M, N, gen=g)
    x = rnd_bf16(M, K, gen=g)
    sp = split if split else M
    ng = 2 if split else 1
    init = torch.randn(ng, N, K, generator=g)
    dw = init.clone().to(DEV)
    db_fused = torch.zeros(ng, N, device=DEV)
    ops.gemm_tn(bf(dy).to(DEV), bf(x).to(DEV), dw, split=split, db=db_fused)
    ref = init.clone().double()
    ref[0] += dy[:sp].double().t() @ x[:sp].double()
    if split:
        ref[1] += dy[sp:].double().t() @ x[sp:].double()
    assert_close(dw, ref.float(), 1e-4, "wgrad (accumulates into dW)")
    db = torch.zeros(ng, N, device=DEV)
    ops.colsum(bf(dy).to(DEV), db, split=split)
    refb = torch.stack([dy[:sp].double().sum(0)] + ([dy[sp:].double().sum(0)] if split else [])).float()
    assert_close(db, refb, 1e-4, "colsum")
    assert_close(db_fused, refb, 1e-4, "bias gradient fused into the wgrad")


@pytest.mark.parametrize("M,N,K,split", [(4133, 3072, 768, 3850), (6011, 2304, 768, 5614), (4500, 768, 3072, 0), (8421, 768, 768, 8020),
                                         (4096, 512, 512, 4090), (26944, 768, 768, 25664), (5000, 512, 1024, 4999),
                                         (4700, 4096, 1024, 4400)])
@pytest.mark.parametrize("slabs", [True, False])
def test_wgrad_square_tile_kernel(M, N, K, split, slabs, monkeypatch):
    """the 16-wave 256 x 256 kernel (csrc/wgrad.hip, wgrad_sq_kernel) forced onto every shape it divides (SIMVG_WGRAD_SQ=1; by
    default it takes the shapes the 12-wave tiles do not divide, e.g. ViT-L's): ragged last stages, a row-group boundary inside a
    partition, one tile (as many partitions as stages), slabs and the atomic flush, the fused bias gradient; run twice: same bits."""
    ops = _ops()
    monkeypatch.setenv("SIMVG_WGRAD_SQ", "1")
    if not slabs:
        monkeypatch.setenv("SIMVG_WG_SLABS", "0")
    g = torch.Generator().manual_seed(11 * M + N + K)
    dy = rnd_bf16(M, N, gen=g)
    x = rnd_bf16(M, K, gen=g)
    sp = split if split else M
    ng = 2 if split else 1
    init = torch.randn(ng, N, K, generator=g)
    ref = init.clone().double()
    ref[0] += dy[:sp].double().t() @ x[:sp].double()
    if split:
        ref[1] += dy[sp:].double().t() @ x[sp:].double()
    refb = torch.stack([dy[:sp].double().sum(0)] + ([dy[sp:].double().sum(0)] if split else [])).float()
    outs = []
    for _ in range(2):
        dw = init.clone().to(DEV)
        db = torch.zeros(ng, N, device=DEV)
        ops.gemm_tn(bf(dy).to(DEV), bf(x).to(DEV), dw, split=split, db=db)
        assert_close(dw, ref.float(), 1e-4, "wgrad, 256 x 256 tiles")
        assert_close(db, refb, 1e-4, "bias gradient, 256 x 256 tiles")
        outs.append(dw)
    if slabs:
        assert torch.equal(outs[0], outs[1]), "slab path is not reproducible run to run"


@pytest.mark.parametrize("M,N,K,split", [(6011, 3072, 768, 5614), (6011, 768, 768, 5614), (4500, 2304, 768, 0)])
def test_wgrad_slabs_deferred_reduction_and_atomics_agree(M, N, K, split, monkeypatch):
    """the XCD-partitioned kernel's partial sums go to per-partition slabs and a second launch adds them to dW in a fixed order:
    (a) deferred into a WgradReduceBatch (several weights, one launch) == immediate, bit for bit; (b) the same call twice gives
    the same bits (the fp32-atomic flush it replaces did not); (c) SIMVG_WG_SLABS=0 (atomics, the A/B switch) agrees to
    rounding of the summation order."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    dy, x = bf(rnd_bf16(M, N, gen=g)).to(DEV), bf(rnd_bf16(M, K, gen=g)).to(DEV)
    ng = 2 if split else 1
    init = torch.randn(ng, N, K, generator=g).to(DEV)

    def run(defer=None, nbatch=1):
        outs = [(init.clone(), torch.zeros(ng, N, device=DEV)) for _ in range(nbatch)]
        for dw, db in outs:
            ops.gemm_tn(dy, x, dw, split=split, db=db, defer=defer)
        if defer is not None:
            defer.flush()
        torch.cuda.synchronize()
        return outs

    (dw0, db0), = run()
    (dw1, db1), = run()
    assert torch.equal(dw0, dw1), "slab path is not reproducible run to run"      # (db still meets through fp32 atomics)
    batch = ops.WgradReduceBatch()
    for dwb, dbb in run(defer=batch, nbatch=3):
        assert torch.equal(dwb, dw0), "deferred (batched) reduction differs from the immediate one"
    for dwb, dbb in run(defer=batch, nbatch=2):           # the batch's workspace pool is reused after a flush
        assert torch.equal(dwb, dw0)
    monkeypatch.setenv("SIMVG_WG_SLABS", "0")
    (dwa, dba), = run()
    monkeypatch.delenv("SIMVG_WG_SLABS")
    assert_close(dwa, dw0, 2e-5, "atomic flush vs slabs")
    assert_close(dba, db0, 2e-5, "bias gradient, atomic flush vs slabs")


@pytest.mark.parametrize("M,N,K,split", [(6011, 3072, 768, 5614), (6011, 768, 768, 5614), (4500, 2304, 768, 0), (4700, 1024, 4096, 4700)])
def test_wgrad_second_stage_assign_mode(M, N, K, split):
    """`gemm_tn(defer=batch, assign=True)`: the second stage WRITES dW (the arena skips its zero fill, simvg_amd/arena.py): on a
    buffer full of garbage it gives, bit for bit, what the accumulating form gives on zeros; a row group without rows gets zeros."""
    ops = _ops()
    assert ops.gemm_tn_can_assign(M, N, K)
    g = torch.Generator().manual_seed(3 * M + N + K)
    dy, x = bf(rnd_bf16(M, N, gen=g)).to(DEV), bf(rnd_bf16(M, K, gen=g)).to(DEV)
    ng = 2 if split else 1
    batch = ops.WgradReduceBatch()
    dw0, db0 = torch.zeros(ng, N, K, device=DEV), torch.zeros(ng, N, device=DEV)
    ops.gemm_tn(dy, x, dw0, split=split, db=db0, defer=batch)
    batch.flush()
    dw1, db1 = torch.full((ng, N, K), float("nan"), device=DEV), torch.zeros(ng, N, device=DEV)
    ops.gemm_tn(dy, x, dw1, split=split, db=db1, defer=batch, assign=True)
    batch.flush()
    torch.cuda.synchronize()
    assert torch.equal(dw0, dw1)
    if split == M:                       # every row in group 0: group 1's gradient is exactly zero
        assert float(dw1[1].abs().max()) == 0.0
    with pytest.raises((RuntimeError, ValueError)):
        ops.gemm_tn(dy, x, dw1, split=split, db=db1, assign=True)          # no defer


def test_deferred_second_stages_survive_an_overflow_flush_of_the_descriptor_table():
    """More deferred calls than the batch's descriptor table holds (16 weight-gradient / 64 LayerNorm descriptors), of MIXED
    sizes: the table flushes in the middle of a batch and frees the workspace pool.  The workspace of the call that triggers the
    flush must be taken AFTER it (round 5, advisor: it used to be taken before and was then no longer counted as in use, so a
    later call of the same size could be handed the same slabs while their second stage was still pending).  Every result must
    equal the immediate (undeferred) call's, bit for bit, and no two descriptors between two flushes may share a workspace."""
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    shapes = [(4500, 768, 768, 0), (4600, 2304, 768, 4200), (4500, 768, 768, 0)]      # two problems share a slab size
    data = []
    for M, N, K, split in shapes:
        dy, x = bf(rnd_bf16(M, N, gen=g)).to(DEV), bf(rnd_bf16(M, K, gen=g)).to(DEV)
        ng = 2 if split else 1
        ref = torch.zeros(ng, N, K, device=DEV)
        ops.gemm_tn(dy, x, ref, split=split)
        data.append((dy, x, split, ref))
    batch = ops.WgradReduceBatch()
    seen = []
    orig = batch.flush

    def flush():                          # every flush: the slabs of its descriptors are pairwise distinct
        ptrs = [batch.descs[i].slabs for i in range(batch.n)]
        assert len(set(ptrs)) == len(ptrs), "two pending second stages share their slabs"
        seen.append(batch.n)
        orig()
    batch.flush = flush
    outs = []
    for r in range(37):                   # 37 calls against a table of 16: two overflow flushes + the final one
        dy, x, split, ref = data[(r * r + r // 3) % 3]
        dw = torch.zeros_like(ref)
        ops.gemm_tn(dy, x, dw, split=split, defer=batch)
        outs.append((dw, ref))
    batch.flush()
    torch.cuda.synchronize()
    assert seen == [16, 16, 5], seen
    for dw, ref in outs:
        assert torch.equal(dw, ref)
    # the LayerNorm parameter reductions: 70 deferred calls (table of 64), two row counts
    lb = ops.LnReduceBatch()
    cases = []
    for M, D, split in [(2048, 768, 1500), (3000, 768, 0)]:
        x = bf(rnd_bf16(M, D, gen=g)).to(DEV)
        dy = bf(rnd_bf16(M, D, gen=g)).to(DEV)
        ng = 2 if split else 1
        gamma, beta = (1 + 0.1 * torch.randn(ng, D, generator=g)).to(DEV), torch.zeros(ng, D, device=DEV)
        _, _, mean, rstd = ops.ln_fwd(x, gamma, beta, split=split)
        dgr, dbr, dxr = torch.zeros(ng, D, device=DEV), torch.zeros(ng, D, device=DEV), torch.empty(M, D, device=DEV, dtype=LPD())
        ops.ln_bwd(dy, x, mean, rstd, gamma, dgr, dbr, split=split, dx_lp=dxr)
        cases.append((dy, x, mean, rstd, gamma, split, dgr, dbr))
    res = []
    for r in range(70):
        dy, x, mean, rstd, gamma, split, dgr, dbr = cases[(r // 5) % 2]
        dg, db_ = torch.zeros_like(dgr), torch.zeros_like(dbr)
        ops.ln_bwd(dy, x, mean, rstd, gamma, dg, db_, split=split, dx_lp=torch.empty_like(x), defer=lb)
        res.append((dg, db_, dgr, dbr))
    lb.flush()
    torch.cuda.synchronize()
    for dg, db_, dgr, dbr in res:
        assert torch.equal(dg, dgr) and torch.equal(db_, dbr)


# ------------------------------------------------------------------------------------------
# LayerNorm
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,D,split", [(100, 768, 70), (37, 3072, 0), (64, 256, 0), (90, 1024, 33), (50, 128, 20),
                                       (41, 4096, 11)])
@pytest.mark.parametrize("xdtype", ["f32", "bf16"])
def test_layernorm_fwd_bwd(M, D, split, xdtype):
    ops = _ops()
    g = torch.Generator().manual_seed(M * D)
    x = torch.randn(M, D, generator=g) * 2 + 0.5
    if xdtype == "bf16":
        x = x.to(torch.bfloat16).float()
    ng = 2 if split else 1
    gamma = 1 + 0.2 * torch.randn(ng, D, generator=g)
    beta = 0.1 * torch.randn(ng, D, generator=g)
    sp = split if split else M
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y_ref = torch.cat([F.layer_norm(xr[:sp], (D,), gr[0], br[0], 1e-5), F.layer_norm(xr[sp:], (D,), gr[-1], br[-1], 1e-5)], 0)
    xd = x.to(DEV) if xdtype == "f32" else bf(x).to(DEV)
    y, y32, mean, rstd = ops.ln_fwd(xd, gamma.to(DEV), beta.to(DEV), split=split, out_lp=True, out_f32=True)
    assert_close(y32, y_ref, 1e-5 if xdtype == "f32" else 1e-5, "ln fwd f32")
    assert_close(y, y_ref, LPTOL(), "ln fwd bf16")
    # backward
    dy = rnd_bf16(M, D, gen=g)
    y_ref.backward(dy)
    dgamma, dbeta = torch.zeros(ng, D, device=DEV), torch.zeros(ng, D, device=DEV)
    dres = torch.randn(M, D, generator=g)
    rps = (7, 3) if split else (5, 1)
    nsamp = max(math.ceil(sp / rps[0]), math.ceil((M - sp) / rps[1]) if split else 1)
    scale = torch.rand(nsamp, generator=g) + 0.5
    dx_f32 = torch.empty(M, D, device=DEV)
    dx_scaled = torch.empty(M, D, device=DEV, dtype=LPD())
    ops.ln_bwd(bf(dy).to(DEV), xd, mean, rstd, gamma.to(DEV), dgamma, dbeta, split=split, dres=dres.to(DEV),
               dx_f32=dx_f32, dx_scaled=dx_scaled, row_scale=scale.to(DEV), rows_per_sample=rps)
    assert_close(dx_f32, dres + xr.grad, 1e-4, "ln bwd dx (+residual)")
    rows = torch.arange(M)
    samp = torch.where(rows < sp, rows // rps[0], (rows - sp) // rps[1])
    assert_close(dx_scaled, (dres + xr.grad) * scale[samp][:, None], LPTOL(), "ln bwd scaled bf16 copy")
    assert_close(dgamma, gr.grad, 1e-4, "ln bwd dgamma")
    assert_close(dbeta, br.grad, 1e-4, "ln bwd dbeta")
    # bf16 output with fused GELU'
    u = rnd_bf16(M, D, gen=g)
    dxb = torch.empty(M, D, device=DEV, dtype=LPD())
    dgamma.zero_(); dbeta.zero_()
    ops.ln_bwd(bf(dy).to(DEV), xd, mean, rstd, gamma.to(DEV), dgamma, dbeta, split=split, dx_lp=dxb, gelu_u=bf(u).to(DEV))
    ur = u.clone().requires_grad_(True)
    F.gelu(ur).backward(xr.grad)
    assert_close(dxb, ur.grad, LPTOL(), "ln bwd * gelu'")


@pytest.mark.parametrize("M,D,split", [(90, 3072, 61), (37, 256, 0), (33, 4096, 9)])
def test_layernorm_of_recomputed_gelu(M, D, split):
    """ffn_layernorm(gelu(u)) with only the pre-activation u stored: forward recomputes gelu(u), backward recomputes
    gelu(u) AND gelu'(u) from the same u (x == gelu_u) -- reference: torchscale FeedForwardNetwork.forward."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + D)
    u = (torch.randn(M, D, generator=g) * 1.5).to(torch.bfloat16).float()
    ng = 2 if split else 1
    gamma, beta = 1 + 0.2 * torch.randn(ng, D, generator=g), 0.1 * torch.randn(ng, D, generator=g)
    sp = split if split else M
    ur = u.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    a = F.gelu(ur)
    y_ref = torch.cat([F.layer_norm(a[:sp], (D,), gr[0], br[0], 1e-5), F.layer_norm(a[sp:], (D,), gr[-1], br[-1], 1e-5)], 0)
    ud = bf(u).to(DEV)
    y, y32, mean, rstd = ops.ln_fwd(ud, gamma.to(DEV), beta.to(DEV), split=split, out_lp=True, out_f32=True, gelu_in=True)
    assert_close(y32, y_ref, 2e-5, "ln(gelu(u)) fwd")
    dy = rnd_bf16(M, D, gen=g)
    y_ref.backward(dy)
    dgamma, dbeta = torch.zeros(ng, D, device=DEV), torch.zeros(ng, D, device=DEV)
    dxb = torch.empty(M, D, device=DEV, dtype=LPD())
    ops.ln_bwd(bf(dy).to(DEV), ud, mean, rstd, gamma.to(DEV), dgamma, dbeta, split=split, dx_lp=dxb, gelu_u=ud)
    assert_close(dxb, ur.grad, LPTOL(), "d/du of ln(gelu(u))")
    assert_close(dgamma, gr.grad, 1e-4, "dgamma")
    assert_close(dbeta, br.grad, 1e-4, "dbeta")


@pytest.mark.parametrize("M,D,split", [(1500, 3072, 1390), (1037, 4096, 0), (1100, 3072, 0)])
def test_ffn_layernorm_hot_kernels(M, D, split):
    """The instances the encoder actually launches at training sizes (M >= 1024 rows, bf16-only outputs): the GELU-fused
    forward and the two-row-prefetch backward with the two-stage dgamma / dbeta reduction -- ragged row counts and row
    groups included."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + D)
    u = (torch.randn(M, D, generator=g) * 1.5 + 0.3).to(torch.bfloat16).float()
    ng = 2 if split else 1
    gamma, beta = 1 + 0.2 * torch.randn(ng, D, generator=g), 0.1 * torch.randn(ng, D, generator=g)
    sp = split if split else M
    ur = u.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    a = F.gelu(ur)
    y_ref = torch.cat([F.layer_norm(a[:sp], (D,), gr[0], br[0], 1e-5), F.layer_norm(a[sp:], (D,), gr[-1], br[-1], 1e-5)], 0)
    ud = bf(u).to(DEV)
    y, y32, mean, rstd = ops.ln_fwd(ud, gamma.to(DEV), beta.to(DEV), split=split, out_lp=True, out_f32=False, gelu_in=True)
    assert y32 is None
    assert_close(y, y_ref, LPTOL(), "ln(gelu(u)) fwd, bf16")
    ad = a.detach().double()
    assert_close(mean, ad.mean(1).float(), 1e-5, "row mean")
    assert_close(rstd, (ad.var(1, unbiased=False) + 1e-5).rsqrt().float(), 1e-5, "row rstd")
    dy = rnd_bf16(M, D, gen=g)
    y_ref.backward(dy)
    dgamma, dbeta = torch.zeros(ng, D, device=DEV), torch.zeros(ng, D, device=DEV)
    dxb = torch.empty(M, D, device=DEV, dtype=LPD())
    ops.ln_bwd(bf(dy).to(DEV), ud, mean, rstd, gamma.to(DEV), dgamma, dbeta, split=split, dx_lp=dxb, gelu_u=ud)
    assert_close(dxb, ur.grad, LPTOL(), "d/du of ln(gelu(u))")
    assert_close(dgamma, gr.grad, 2e-4, "dgamma")
    assert_close(dbeta, br.grad, 2e-4, "dbeta")


@pytest.mark.parametrize("M,D,split", [(1203, 768, 1100), (77, 1024, 0)])
def test_sub_layernorm_backward_prefetch_kernel(M, D, split):
    """bf16 x, bf16 dy, bf16 dx only (the attention sub-LayerNorm instance of the prefetching backward)"""
    ops = _ops()
    g = torch.Generator().manual_seed(M)
    x = rnd_bf16(M, D, gen=g, scale=2.0)
    ng = 2 if split else 1
    gamma, beta = 1 + 0.2 * torch.randn(ng, D, generator=g), 0.1 * torch.randn(ng, D, generator=g)
    sp = split if split else M
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y_ref = torch.cat([F.layer_norm(xr[:sp], (D,), gr[0], br[0], 1e-5), F.layer_norm(xr[sp:], (D,), gr[-1], br[-1], 1e-5)], 0)
    xd = bf(x).to(DEV)
    _, _, mean, rstd = ops.ln_fwd(xd, gamma.to(DEV), beta.to(DEV), split=split, out_lp=True, out_f32=False)
    dy = rnd_bf16(M, D, gen=g)
    y_ref.backward(dy)
    dgamma, dbeta = torch.zeros(ng, D, device=DEV), torch.zeros(ng, D, device=DEV)
    dxb = torch.empty(M, D, device=DEV, dtype=LPD())
    ops.ln_bwd(bf(dy).to(DEV), xd, mean, rstd, gamma.to(DEV), dgamma, dbeta, split=split, dx_lp=dxb)
    assert_close(dxb, xr.grad, LPTOL(), "dx")
    assert_close(dgamma, gr.grad, 2e-4, "dgamma")
    assert_close(dbeta, br.grad, 2e-4, "dbeta")


def test_layernorm_backward_deferred_second_stage_is_bit_identical():
    """`ln_bwd(..., defer=batch)` + one `simvg_ln_param_reduce_batched` launch for SEVERAL calls == the per-call second stage:
    same partial sums, same order -> bit-identical dgamma / dbeta, for the three two-stage instances of a training step (the
    residual-stream backward with fp32 x and dres, the attention sub-LayerNorm, the wide FFN LayerNorm with GELU) and widths."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    M, split = 1500, 1390
    cases = []
    for D, kind in ((768, "res"), (768, "sub"), (3072, "ffn"), (1024, "sub"), (768, "res")):
        gamma = (1 + 0.2 * torch.randn(2, D, generator=g)).to(DEV)
        beta = (0.1 * torch.randn(2, D, generator=g)).to(DEV)
        dy = bf(rnd_bf16(M, D, gen=g)).to(DEV)
        if kind == "res":
            x = torch.randn(M, D, generator=g).to(DEV)
            _, _, mean, rstd = ops.ln_fwd(x, gamma, beta, split=split, out_lp=True, out_f32=False)
            kw = dict(dres=torch.randn(M, D, generator=g).to(DEV), dx_f32=torch.empty(M, D, device=DEV),
                      dx_scaled=torch.empty(M, D, device=DEV, dtype=LPD()))
        elif kind == "sub":
            x = bf(rnd_bf16(M, D, gen=g, scale=2.0)).to(DEV)
            _, _, mean, rstd = ops.ln_fwd(x, gamma, beta, split=split, out_lp=True, out_f32=False)
            kw = dict(dx_lp=torch.empty(M, D, device=DEV, dtype=LPD()))
        else:
            x = bf(rnd_bf16(M, D, gen=g, scale=1.5)).to(DEV)
            _, _, mean, rstd = ops.ln_fwd(x, gamma, beta, split=split, out_lp=True, out_f32=False, gelu_in=True)
            kw = dict(dx_lp=torch.empty(M, D, device=DEV, dtype=LPD()), gelu_u=x)
        cases.append((dy, x, mean, rstd, gamma, kw))
    ref = []
    for dy, x, mean, rstd, gamma, kw in cases:
        dg, db = torch.zeros_like(gamma), torch.zeros_like(gamma)
        ops.ln_bwd(dy, x, mean, rstd, gamma, dg, db, split=split, param_scale=0.5, **kw)
        ref.append((dg, db, {k: v.clone() for k, v in kw.items() if k.startswith("dx")}))
    batch = ops.LnReduceBatch()
    got = []
    for dy, x, mean, rstd, gamma, kw in cases:
        dg, db = torch.zeros_like(gamma), torch.zeros_like(gamma)
        ops.ln_bwd(dy, x, mean, rstd, gamma, dg, db, split=split, param_scale=0.5, defer=batch, **kw)
        got.append((dg, db))
    assert batch.n == len(cases)
    assert float(got[0][0].abs().max()) == 0.0            # nothing reduced before the flush
    batch.flush()
    assert batch.n == 0
    for i, ((dg, db), (rg, rb, dxs)) in enumerate(zip(got, ref)):
        assert torch.equal(dg, rg) and torch.equal(db, rb), i
        assert float(rg.abs().max()) > 0
        for k, v in dxs.items():
            assert torch.equal(cases[i][5][k], v), (i, k)


# ------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------
def _mm_rows(B, Nv, Nt):
    """modality-major row index of token (b, t)."""
    idx = torch.zeros(B, Nv + Nt, dtype=torch.long)
    for b in range(B):
        idx[b, :Nv] = b * Nv + torch.arange(Nv)
        idx[b, Nv:] = B * Nv + b * Nt + torch.arange(Nt)
    return idx


# N <= 448: K / V resident in LDS (421 = the path's compile-time geometry); N > 448: streamed blocks + online softmax
# 27 key tiles with >= 24 of them vision keys (the path's geometry, 421 tokens; 424 / 430 as well): the backward runs in one pass
# (csrc/attention_bwd1.hip) unless SIMVG_ATTN_BWD1=0 selects the dq + dkv kernels
@pytest.mark.parametrize("B,H,Nv,Nt", [(2, 2, 17, 20), (2, 12, 401, 20), (1, 3, 50, 0), (3, 2, 100, 7), (2, 2, 500, 13),
                                       (1, 2, 1601, 20), (2, 1, 512, 0), (3, 16, 401, 20), (1, 4, 412, 12), (2, 3, 410, 20),
                                       (24, 12, 401, 20)])       # 288 workgroups: more than one residency round (the forward's L2 touches run)
@pytest.mark.parametrize("one_pass", [True, False])
def test_attention_fwd_bwd(B, H, Nv, Nt, one_pass, monkeypatch):
    if not one_pass:
        if (Nv + Nt + 15) // 16 != 27:
            pytest.skip("the switch only matters for the 27-tile geometry")
        monkeypatch.setenv("SIMVG_ATTN_BWD1", "0")
    ops = _ops()
    d, N = 64, Nv + Nt
    D = H * d
    g = torch.Generator().manual_seed(B * 1000 + N)
    qkv_tok = rnd_bf16(B, N, 3 * D, gen=g)                        # token-major reference layout
    pad = torch.zeros(B, max(Nt, 1), dtype=torch.uint8)
    for b in range(B):
        if Nt:
            pad[b, Nt - (3 + 4 * b) % Nt:] = 1   # every sample has a different number of padded text keys
    idx = _mm_rows(B, Nv, Nt)
    qkv_mm = torch.zeros(B * N, 3 * D)
    qkv_mm[idx.reshape(-1)] = qkv_tok.reshape(B * N, 3 * D)
    # reference (torchscale MultiheadAttention math, fp32)
    t = qkv_tok.clone().requires_grad_(True)
    q, k, v = t.split(D, dim=-1)
    q = q.view(B, N, H, d).transpose(1, 2) * d ** -0.5
    k = k.view(B, N, H, d).transpose(1, 2)
    v = v.view(B, N, H, d).transpose(1, 2)
    w = q @ k.transpose(-1, -2)
    kpm = torch.cat([torch.zeros(B, Nv, dtype=torch.bool), pad[:, :Nt].bool()], 1)
    w = w.masked_fill(kpm[:, None, None, :], float("-inf"))
    p = torch.softmax(w, -1)
    o_ref = (p @ v).transpose(1, 2).reshape(B, N, D)
    qd = bf(qkv_mm).to(DEV)
    padd = pad[:, :Nt].contiguous().to(DEV) if Nt else None
    out, lse = ops.attn_fwd(qd, B, H, Nv, Nt, pad=padd)
    out_tok = out.float().cpu()[idx.reshape(-1)].view(B, N, D)
    assert_close(out_tok, o_ref, LPTOL(1.5), "attention fwd")
    lse_ref = torch.logsumexp(w, -1).reshape(B * H, N)
    assert_close(lse, lse_ref, 1e-3, "attention lse")
    if (N + 15) // 16 == 27 and Nv // 16 >= 25 and one_pass:
        # the measurement entry point: the forward kernel's QK^T contraction alone (row maxima of the scaled, masked scores)
        assert_close(ops.attn_qk_probe(qd, B, H, Nv, Nt, pad=padd), w.detach().max(-1)[0].reshape(B * H, N), 1e-3, "QK^T probe: row maxima")
    # backward
    do_tok = rnd_bf16(B, N, D, gen=g)
    o_ref.backward(do_tok)
    do_mm = torch.zeros(B * N, D)
    do_mm[idx.reshape(-1)] = do_tok.reshape(B * N, D)
    dqkv = ops.attn_bwd(qd, out, bf(do_mm).to(DEV), lse, B, H, Nv, Nt, pad=padd)
    dq_tok = dqkv.float().cpu()[idx.reshape(-1)].view(B, N, 3 * D)
    for name, sl in [("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))]:
        assert_close(dq_tok[..., sl], t.grad[..., sl], LPTOL(2.0), "attention bwd " + name)


# ------------------------------------------------------------------------------------------
# embedding stage
# ------------------------------------------------------------------------------------------
def test_im2col_matches_conv():
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    B, S, P, D = 2, 128, 32, 64
    img = rnd_bf16(B, 3, S, S, gen=g)
    w = rnd_bf16(D, 3, P, P, scale=0.02, gen=g)
    cols = ops.im2col(img.to(DEV), P)
    y = ops.gemm_nt(cols, bf(w.view(D, -1)).to(DEV), out_dtype=torch.float32)
    ref = F.conv2d(img, w, stride=P).flatten(2).transpose(1, 2).reshape(B * (S // P) ** 2, D)
    assert_close(y, ref, 1e-3, "patch embed = im2col + GEMM")


def test_embed_fwd_bwd():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    B, np_, T, D, V = 3, 16, 20, 128, 1000
    patch = torch.randn(B * np_, D, generator=g)
    cls = torch.randn(D, generator=g)
    posA = torch.randn(np_ + 3, D, generator=g)
    posB = torch.randn(1024, D, generator=g)
    table = torch.randn(V, D, generator=g)
    ids = torch.randint(0, V, (B, T), generator=g)
    ids[:, 0] = 0
    pad = torch.zeros(B, T, dtype=torch.uint8)
    pad[0, 5:] = 1
    pad[2, 12:] = 1
    leaves = [t.clone().requires_grad_(True) for t in (patch, cls, posA, posB, table)]
    pr, cr, ar, br_, tr = leaves
    x1 = torch.cat([cr.view(1, 1, D).expand(B, 1, D), pr.view(B, np_, D)], 1) + ar[2:2 + np_ + 1][None]
    x2 = (F.embedding(ids, tr) + br_[2:2 + T][None]) * (1 - pad.float())[..., None]
    x_ref = torch.cat([x1.reshape(-1, D), x2.reshape(-1, D)], 0)
    x = ops.embed_fwd(patch.to(DEV), cls.to(DEV), posA.to(DEV), posB.to(DEV), table.to(DEV), ids.to(DEV), pad.to(DEV),
                      B, np_, T)
    assert_close(x, x_ref, 1e-6, "embed fwd")
    dx = torch.randn(B * (np_ + 1 + T), D, generator=g)
    x_ref.backward(dx)
    dpatch = torch.empty(B * np_, D, device=DEV, dtype=LPD())
    dcls, dposA, dposB, dtext = (torch.zeros(s, device=DEV) for s in [(D,), (np_ + 3, D), (1024, D), (V, D)])
    ops.embed_bwd(dx.to(DEV), dpatch, dcls, dposA, dposB, dtext, ids.to(DEV), pad.to(DEV), B, np_, T)
    assert_close(dpatch, pr.grad, LPTOL(), "embed bwd dpatch")
    assert_close(dcls, cr.grad, 1e-5, "embed bwd dcls")
    assert_close(dposA, ar.grad, 1e-5, "embed bwd dposA")
    assert_close(dposB, br_.grad, 1e-5, "embed bwd dposB")
    assert_close(dtext, tr.grad, 1e-5, "embed bwd dtext")


def test_weight_prep():
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    mats = [torch.randn(r, c, generator=g).to(DEV) for r, c in [(70, 33), (128, 256), (5, 300), (132, 200), (768, 3072)]]
    entries = []
    for i, m in enumerate(mats):
        dst = torch.empty_like(m, dtype=LPD()) if i != 2 else None
        dst_t = torch.empty(m.shape[1], m.shape[0], device=DEV, dtype=LPD()) if i != 0 else None
        entries.append((m, dst, dst_t))
    wp = ops.WeightPrep(entries, DEV)
    wp.run()
    torch.cuda.synchronize()
    for m, dst, dst_t in entries:
        if dst is not None:
            assert torch.equal(dst, m.to(LPD()))
        if dst_t is not None:
            assert torch.equal(dst_t, m.t().contiguous().to(LPD()))


def test_weight_prep_writes_hi_lo_rows():
    """`simvg_weight_prep` with split_shift (round 6, `BEIT3.precise_training`): dst rows are [lo * 2^11 | hi] -- bit for bit what
    `hip_ops.split_weight` (torch arithmetic: hi = the 16-bit rounding of w, lo = the 16-bit rounding of (w - hi) * 2^11) builds for
    forward_test; the right half IS the plain 16-bit copy; the transposed copy is untouched by the mode; ragged shapes take the
    element-wise path."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    mats = [(torch.randn(r, c, generator=g) * sc).to(DEV) for r, c, sc in [(768, 768, 0.03), (132, 200, 1.0), (70, 33, 0.5), (2304, 768, 0.02)]]
    entries = []
    for i, m in enumerate(mats):
        dst = torch.full((m.shape[0], 2 * m.shape[1]), float("nan"), device=DEV, dtype=LPD())
        dst_t = torch.empty(m.shape[1], m.shape[0], device=DEV, dtype=LPD()) if i != 2 else None
        entries.append((m, dst, dst_t, ops.SPLIT_SHIFT))
    plain = torch.empty_like(mats[0], dtype=LPD())
    entries.append((mats[0], plain, None))                    # a plain entry beside them in the same launch
    ops.WeightPrep(entries, DEV).run()
    torch.cuda.synchronize()
    for m, dst, dst_t, _ in entries[:-1]:
        K = m.shape[1]
        assert torch.equal(dst, ops.split_weight(m)), m.shape
        assert torch.equal(dst[:, K:], m.to(LPD()))
        if dst_t is not None:
            assert torch.equal(dst_t, m.t().contiguous().to(LPD()))
        # hi + lo carries the weight to ~2^-22 relative (fp16 build; bf16: 2^-16)
        rec = dst[:, K:].double() + dst[:, :K].double() * 2.0 ** -ops.SPLIT_SHIFT
        rel = float((rec - m.double()).abs().max() / m.double().abs().max())
        assert rel <= (2e-6 if LPD() == torch.float16 else 1e-4), rel
    assert torch.equal(plain, mats[0].to(LPD()))


def test_dropout_multipliers_philox():
    """csrc/rng.hip: multipliers are 0 or 1 / keep with the right frequency, a pure function of (key, index) -- the key of
    every call is drawn from torch's CPU generator, so `torch.manual_seed` reproduces the sequence of masks -- and the
    per-segment table gives every encoder layer its own DropPath rate (torchscale DropPath: Bernoulli(1 - p) / (1 - p))."""
    ops = _ops()
    torch.manual_seed(77)
    n = 1 << 20
    a = ops.dropout_mult(n, DEV, keep=0.9)
    b = ops.dropout_mult(n, DEV, keep=0.9)
    torch.manual_seed(77)
    a2 = ops.dropout_mult(n, DEV, keep=0.9)
    b2 = ops.dropout_mult(n + 3, DEV, keep=0.9)            # a ragged length: the shared prefix is the same stream
    assert torch.equal(a, a2) and torch.equal(b, b2[:n]) and not torch.equal(a, b)
    vals = torch.unique(a)
    assert vals.numel() == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1 / 0.9) < 1e-6
    frac = float((a > 0).float().mean())
    assert abs(frac - 0.9) < 4 * (0.9 * 0.1 / n) ** 0.5 + 1e-4, frac                    # 4 sigma
    assert abs(float(a.mean()) - 1.0) < 2e-3                                           # unbiased
    # no correlation between neighbours / between the four words of a counter
    k = (a > 0).float()
    for lag in (1, 2, 3, 4, 64):
        c = float(((k[:-lag] - 0.9) * (k[lag:] - 0.9)).mean()) / (0.9 * 0.1)
        assert abs(c) < 6e-3, (lag, c)
    # DropPath table: L layers x 2 draws x B samples, one keep probability per layer
    L, B = 12, 4096
    keep = torch.linspace(1.0, 0.9, L).to(DEV)
    m = ops.dropout_mult(L * 2 * B, DEV, keep_seg=keep, seg=2 * B).view(L, 2 * B)
    assert bool((m[0] == 1.0).all())                                                     # keep = 1: never dropped
    for i in range(1, L):
        kp = float(keep[i])
        f = float((m[i] > 0).float().mean())
        assert abs(f - kp) < 4 * (kp * (1 - kp) / (2 * B)) ** 0.5 + 1e-3, (i, f, kp)
        nz = m[i][m[i] > 0]
        assert abs(float(nz[0]) - 1 / kp) < 1e-5
