"""torch.autograd.Function adapters over the HIP head kernels (libsimvg_hip.so).

Each Function is the (forward, dgrad, wgrad) triple of ONE reference leaf op, so the head's dataflow
(`tgqs_kd_detr_head.py:375-454` of the reference) can be written once in Python while every FLOP runs in a
hand-written gfx950 kernel.  PyTorch autograd is used as the tape only.
"""
import torch

from ... import hip_ops as ops

_ones_cache = {}


def _ones(n, device):
    t = _ones_cache.get((n, device))
    if t is None:
        t = torch.ones(n, device=device)
        _ones_cache[(n, device)] = t
    return t


class LinearF32(torch.autograd.Function):
    """y = x W^T + b (+ReLU), exact fp32 on MFMA f32 (small M).  x may be any 2-D strided view."""

    @staticmethod
    def forward(ctx, x, W, b, relu):
        M, K = x.shape
        N = W.shape[0]
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        ops.gemm_f32(x, x.stride(0), x.stride(1), W, 1, W.stride(0), y, M, N, K, bias=b, act=2 if relu else 0)
        ctx.save_for_backward(x, W, y if relu else None)
        ctx.relu, ctx.has_b = relu, b is not None
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:          # reachable in the graph but without a gradient (input_cls_proj under a decoder-only head): no
            return None, None, None, None        # gradient for the parameters either, as in the reference -- not zeros
        x, W, y = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.relu:
            dy = torch.ops.aten.threshold_backward(dy, y, 0.0)
        M, K = x.shape
        N = W.shape[0]
        dx = dW = db = None
        probs = []                        # weight, bias and input gradient: independent contractions of dy, ONE launch
        if ctx.needs_input_grad[1]:
            dW = torch.empty(N, K, device=dy.device, dtype=torch.float32)
            probs.append(ops.gp(dy, 1, N, x, x.stride(0), x.stride(1), dW, N, K, M))                     # dW = dy^T x
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = torch.empty(1, N, device=dy.device, dtype=torch.float32)
            probs.append(ops.gp(_ones(M, dy.device), 0, 1, dy, N, 1, db, 1, N, M))                       # db = 1^T dy
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, device=dy.device, dtype=torch.float32)
            probs.append(ops.gp(dy, N, 1, W, W.stride(0), 1, dx, M, K, N))                               # dx = dy W
        if probs:
            ops.gemm_f32_group(probs)
        return dx, dW, None if db is None else db.view(N), None


class TextFilt(torch.autograd.Function):
    """filt [B, E] (the "filtered" text feature the TGQG adds to the query embedding, tgqs_kd_detr_head.py:385-388, with an int64
    mask: quirk Q1) and the uint8 key-padding mask of the text rows, one launch; backward one launch."""

    @staticmethod
    def forward(ctx, text, mask, B, T):
        text = text.contiguous()
        lib = ops._lib.load()
        filt = torch.empty(B, text.shape[1], device=text.device, dtype=torch.float32)
        kpm = torch.empty(B, T, device=text.device, dtype=torch.uint8)
        ops._lib.check(lib.simvg_text_filt_fwd(text.data_ptr(), mask.data_ptr(), filt.data_ptr(), kpm.data_ptr(), B, T, ops._stream()),
                       "simvg_text_filt_fwd")
        ctx.save_for_backward(text, mask)
        ctx.geo = (B, T)
        ctx.mark_non_differentiable(kpm)
        return filt, kpm

    @staticmethod
    def backward(ctx, dfilt, _):
        text, mask = ctx.saved_tensors
        B, T = ctx.geo
        dtext = torch.empty_like(text)
        lib = ops._lib.load()
        ops._lib.check(lib.simvg_text_filt_bwd(text.data_ptr(), mask.data_ptr(), dfilt.contiguous().data_ptr(), dtext.data_ptr(), B, T,
                                               ops._stream()), "simvg_text_filt_bwd")
        return dtext, None, None, None


class QueryMix(torch.autograd.Function):
    """query_embed = g + filt + qe, tok = query_embed + cls (tgqs_kd_detr_head.py:399-411, Q5): one launch; the backward (two small
    launches) takes the gradients of both outputs and returns those of g, filt, qe, cls."""

    @staticmethod
    def forward(ctx, g, filt, qe, cls, B, R):
        g, filt, qe, cls = g.contiguous(), filt.contiguous(), qe.contiguous(), cls.contiguous()
        qeo, tok = torch.empty_like(g), torch.empty_like(g)
        lib = ops._lib.load()
        ops._lib.check(lib.simvg_query_mix_fwd(g.data_ptr(), filt.data_ptr(), qe.data_ptr(), cls.data_ptr(), qeo.data_ptr(), tok.data_ptr(),
                                               B, R, ops._stream()), "simvg_query_mix_fwd")
        ctx.geo = (B, R, g.shape[1])
        ctx.set_materialize_grads(False)
        return qeo, tok

    @staticmethod
    def backward(ctx, dqeo, dtok):
        B, R, E = ctx.geo
        ref = dqeo if dqeo is not None else dtok
        dqeo = None if dqeo is None else dqeo.contiguous()
        dtok = None if dtok is None else dtok.contiguous()
        buf = torch.empty((B * R + 2 * B + R) * E, device=ref.device, dtype=torch.float32)
        dg, dfilt, dcls, dqe = buf[:B * R * E], buf[B * R * E:(B * R + B) * E], buf[(B * R + B) * E:(B * R + 2 * B) * E], buf[(B * R + 2 * B) * E:]
        lib = ops._lib.load()
        ops._lib.check(lib.simvg_query_mix_bwd(None if dqeo is None else dqeo.data_ptr(), None if dtok is None else dtok.data_ptr(),
                                               dg.data_ptr(), dfilt.data_ptr(), dcls.data_ptr(), dqe.data_ptr(), B, R, ops._stream()),
                       "simvg_query_mix_bwd")
        # (no gradient reaches `tok` when the head has no token branch: `cls` then gets None, as in the reference, not zeros)
        return dg.view(B * R, E), dfilt.view(B, E), dqe.view(R, E), (dcls.view(B, E) if dtok is not None else None), None, None


class PredHeadFn(torch.autograd.Function):
    """The prediction heads on [M, E] query rows as one autograd node: optional leading Linear (the token branch's `mlp.layers.0`,
    tgqs_kd_detr_head.py:411-413), class Linear, and the 3-layer box MLP (Linear, ReLU, Linear, ReLU, Linear; heads/utils.py:39-46)
    + sigmoid (:415-420, :427-428).  Forward: 3 (4) launches -- {class logits, first box Linear + ReLU} share one, the sigmoid is the
    last GEMM's epilogue --; backward: 3 (4) grouped launches + the sigmoid's derivative -- each stage's weight, bias and input
    gradient are one launch, the ReLU gates ride on the input gradient's epilogue (gate = the saved activation), the class branch's
    input gradient joins the box branch's as the epilogue's addend -- instead of 5 (6) / 16 (19).
    Returns (logits [M, C], boxes [M, 4], the leading Linear's output [M, E] | None)."""

    @staticmethod
    def forward(ctx, x, Wm, bm, Wc, bc, W0, b0, W1, b1, W2, b2):
        x = x.contiguous()
        M, E = x.shape
        dev = x.device
        xm = x
        if Wm is not None:
            xm = _f32(M, Wm.shape[0], device=dev)
            ops.gemm_f32(x, E, 1, Wm, 1, Wm.stride(0), xm, M, Wm.shape[0], E, bias=bm)
        C, F0, F1 = Wc.shape[0], W0.shape[0], W1.shape[0]
        logits, h0, h1, boxes = _f32(M, C, device=dev), _f32(M, F0, device=dev), _f32(M, F1, device=dev), _f32(M, W2.shape[0], device=dev)
        Em = xm.shape[1]
        ops.gemm_f32_group([ops.gp(xm, Em, 1, Wc, 1, Wc.stride(0), logits, M, C, Em, bias=bc),
                            ops.gp(xm, Em, 1, W0, 1, W0.stride(0), h0, M, F0, Em, bias=b0, act=2)])
        ops.gemm_f32(h0, F0, 1, W1, 1, W1.stride(0), h1, M, F1, F0, bias=b1, act=2)
        ops.gemm_f32(h1, F1, 1, W2, 1, W2.stride(0), boxes, M, W2.shape[0], F1, bias=b2, act=3)
        ctx.save_for_backward(x, xm, h0, h1, boxes, Wm, Wc, W0, W1, W2)
        ctx.set_materialize_grads(False)
        return logits, boxes, (xm if Wm is not None else None)

    @staticmethod
    def backward(ctx, dlogits, dboxes, dxm):
        x, xm, h0, h1, boxes, Wm, Wc, W0, W1, W2 = ctx.saved_tensors
        M, E = x.shape
        dev = x.device
        Em, C, F0, F1, NB = xm.shape[1], Wc.shape[0], W0.shape[0], W1.shape[0], W2.shape[0]
        ones = _ones(M, dev)
        z = lambda t, shape: torch.zeros(shape, device=dev, dtype=torch.float32) if t is None else t.contiguous()
        dlogits, dboxes = z(dlogits, (M, C)), z(dboxes, (M, NB))
        dpre2 = torch.ops.aten.sigmoid_backward(dboxes, boxes)
        dWc, dbc, dW0, db0 = _f32(C, Em, device=dev), _f32(1, C, device=dev), _f32(F0, Em, device=dev), _f32(1, F0, device=dev)
        dW1, db1, dW2, db2 = _f32(F1, F0, device=dev), _f32(1, F1, device=dev), _f32(NB, F1, device=dev), _f32(1, NB, device=dev)
        dh1, dh0, dxc, dxm_ = _f32(M, F1, device=dev), _f32(M, F0, device=dev), _f32(M, Em, device=dev), _f32(M, Em, device=dev)
        ops.gemm_f32_group([ops.gp(dpre2, 1, NB, h1, F1, 1, dW2, NB, F1, M),
                            ops.gp(ones, 0, 1, dpre2, NB, 1, db2, 1, NB, M),
                            ops.gp(dpre2, NB, 1, W2, W2.stride(0), 1, dh1, M, F1, NB, gate=h1),         # (dpre2 W2) where the ReLU was open
                            ops.gp(dlogits, C, 1, Wc, Wc.stride(0), 1, dxc, M, Em, C),                  # the class branch's share of d(xm)
                            ops.gp(dlogits, 1, C, xm, Em, 1, dWc, C, Em, M),
                            ops.gp(ones, 0, 1, dlogits, C, 1, dbc, 1, C, M)])
        ops.gemm_f32_group([ops.gp(dh1, 1, F1, h0, F0, 1, dW1, F1, F0, M),
                            ops.gp(ones, 0, 1, dh1, F1, 1, db1, 1, F1, M),
                            ops.gp(dh1, F1, 1, W1, W1.stride(0), 1, dh0, M, F0, F1, gate=h0)])
        addend = dxc if dxm is None else dxc + dxm.contiguous()
        ops.gemm_f32_group([ops.gp(dh0, 1, F0, xm, Em, 1, dW0, F0, Em, M),
                            ops.gp(ones, 0, 1, dh0, F0, 1, db0, 1, F0, M),
                            ops.gp(dh0, F0, 1, W0, W0.stride(0), 1, dxm_, M, Em, F0, addend=addend, addend_rows=M)])
        dx, dWm, dbm = dxm_, None, None
        if Wm is not None:
            dWm, dbm, dx = _f32(Wm.shape[0], E, device=dev), _f32(1, Wm.shape[0], device=dev), _f32(M, E, device=dev)
            ops.gemm_f32_group([ops.gp(dxm_, 1, Em, x, E, 1, dWm, Em, E, M),
                                ops.gp(ones, 0, 1, dxm_, Em, 1, dbm, 1, Em, M),
                                ops.gp(dxm_, Em, 1, Wm, Wm.stride(0), 1, dx, M, E, Em)])
            dbm = dbm.view(-1)
        return (dx if ctx.needs_input_grad[0] else None, dWm, dbm, dWc, dbc.view(C), dW0, db0.view(F0), dW1, db1.view(F1), dW2, db2.view(NB))


class LinearLP(torch.autograd.Function):
    """y = x W^T + b on the 16-bit MFMA GEMM for the B*(1+HW) memory rows.  x [M,K] in the library's 16-bit format
    (hip_ops.LP()); W fp32 master [N,K]; the 16-bit copies (plain and transposed) are refreshed by the caller's
    weight-prep.  Gradients that cross autograd are true fp32 gradients: the backward rounds dy * S to 16 bits
    (S = hip_ops.grad_scale(), fp16's exponent range) and the GEMMs remove the scale again."""

    @staticmethod
    def forward(ctx, x, W, b, w_lp, wT_lp, out_lp, holder=None, xholder=None):
        y = ops.gemm_nt(x, w_lp, bias=b, out_dtype=ops.LP() if out_lp else torch.float32)
        ctx.save_for_backward(x, wT_lp)
        ctx.shapeW = W.shape
        # holder (the dict of `SharedMemoryGrad.join`): the fp32 gradient of a 16-bit output is handed over on the side -- through the
        # graph's edge autograd would round it to the output's 16-bit dtype (unscaled!) and this node would widen it again
        ctx.holder = holder
        ctx.xholder = xholder        # the same for the 16-bit INPUT: its fp32 gradient is left in xholder["dx32"] for the producer
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wT = ctx.saved_tensors
        S = ops.grad_scale()
        side = ctx.holder.pop("dy32", None) if ctx.holder is not None else None
        if side is not None:
            dy = side if dy is None else side + dy.float()
        if dy is None:
            return None, None, None, None, None, None, None, None
        dyb = ops.cast_lp(dy.float().contiguous(), scale=S)
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            # x is the leading rows of a larger row-major buffer (the vision rows of the encoder output, text rows behind
            # them): the gradient is written into the leading rows of an fp32 buffer of the WHOLE extent, so that
            # SplitEncoderOutput.backward can fill in the remaining rows instead of copying these 83 MB
            M, K = x.shape
            rows_total = x.untyped_storage().nbytes() // (x.element_size() * K) if x.is_contiguous() and x.storage_offset() == 0 else M
            full = torch.empty(max(rows_total, M), K, device=dy.device, dtype=torch.float32)
            dx = ops.gemm_nt(dyb, wT, out=full[:M], alpha=1.0 / S)                       # [M,K] fp32
        if ctx.needs_input_grad[1]:
            dW = torch.zeros(ctx.shapeW, device=dy.device, dtype=torch.float32)
            db = torch.zeros(ctx.shapeW[0], device=dy.device, dtype=torch.float32) if ctx.needs_input_grad[2] else None
            ops.gemm_tn(dyb, x, dW.view(ctx.shapeW[0], -1), db=db, out_scale=1.0 / S)
        elif ctx.needs_input_grad[2]:
            db = torch.zeros(ctx.shapeW[0], device=dy.device, dtype=torch.float32)
            ops.colsum(dyb, db)
            db.mul_(1.0 / S)
        if dx is not None and ctx.xholder is not None and x.dtype != torch.float32:
            ctx.xholder["dx32"] = dx         # (through the edge autograd would round these 83 MB to x's 16-bit dtype, unscaled)
            dx = None
        return dx, dW, db, None, None, None, None, None


class LayerNormF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, eps):
        x = x.contiguous()
        _, y, mean, rstd = ops.ln_fwd(x, g, b, eps=eps, out_lp=False, out_f32=True)
        ctx.save_for_backward(x, g, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        dgb = torch.zeros(2, g.numel(), device=g.device, dtype=torch.float32)      # the kernel accumulates: one fill for both
        dg, db = dgb[0].view_as(g), dgb[1].view_as(g)
        dx = torch.empty_like(x)
        ops.ln_bwd(dy, x, mean, rstd, g, dg, db, dx_f32=dx)
        return dx, dg, db, None


class SmallAttention(torch.autograd.Function):
    """nn.MultiheadAttention core: per (sample, head) softmax(q k^T / sqrt(32) + kpm) [* dropout] v.
    q [B*Lq,E]; k, v: views into [B*kv_rows, *] buffers whose first used row is the view's first row."""

    @staticmethod
    def forward(ctx, q, k, v, B, H, Lq, Lk, kpm, drop, kv_rows):
        out, P = ops.attn_small_fwd(q, k, v, B, H, Lq, Lk, kpm=kpm, drop=drop, kv_rows=kv_rows)
        ctx.save_for_backward(q, k, v, P, kpm, drop)
        ctx.geo = (B, H, Lq, Lk, kv_rows)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, P, kpm, drop = ctx.saved_tensors
        B, H, Lq, Lk, kv_rows = ctx.geo
        dq = torch.empty_like(q)
        # the kernel writes rows b*kv_rows + [0, Lk) of dk / dv; only the rows it does not own (the CLS key between two
        # samples' patch keys) are cleared -- not a 26 MB fill per cross-attention layer
        dk, dv = torch.empty_like(k), torch.empty_like(v)
        stride_rows = kv_rows if kv_rows else Lk
        for j in range(Lk, stride_rows):
            dk[j::stride_rows].zero_()
            dv[j::stride_rows].zero_()
        ops.attn_small_bwd(q, k, v, P, dout.contiguous(), dq, dk, dv, B, H, Lq, Lk, kpm=kpm, drop=drop, kv_rows=kv_rows)
        return dq, dk, dv, None, None, None, None, None, None, None


def _f32(*shape, device):
    return torch.empty(*shape, device=device, dtype=torch.float32)


class LayerCfg:
    """non-tensor arguments of `DecoderLayerFn` (geometry, cross-attention flavour and its gradient-free operands).
    kind "text": the source rows are fp32 [B*Lk, E] (keys = rows + pos, values = rows: the TGQG layers, pos = the 1-D sine table);
    kind "mem": the image memory [B*Nv, E] (16-bit or fp32), the keys are rows 1 .. Nv-1 of each sample (+ pos, the 2-D sine rows
    [HW, E] shared by the batch or [B, HW, E]); kpm [B, Lk] uint8 masks keys.
    mem_grad: the `SharedMemoryGrad` holder of the memory rows, or None (the layer returns its own d(source rows))"""

    def __init__(self, B, H, nq, kind, Lk, kpm=None, pos=None, Nv=0, p_attn=0.0, p_ffn=0.0, training=False, mask_fn=None,
                 mem_grad=None, wb=None, wbT=None):
        self.B, self.H, self.nq, self.kind, self.Lk = B, H, nq, kind, Lk
        self.kpm, self.pos, self.Nv = kpm, pos, Nv
        self.wb, self.wbT = wb, wbT          # 16-bit K|V weights of the unfused layer (DecoderLayerUnfusedFn) only
        self.p_attn, self.p_ffn, self.training, self.mask_fn = p_attn, p_ffn, training, mask_fn
        self.mem_grad = mem_grad


class SharedMemoryGrad(torch.autograd.Function):
    """The image memory is read by every decoder layer; autograd would add the layers' [B*Nv, E] fp32 gradients pairwise
    (two 26 MB passes for three layers).  `mem, holder = SharedMemoryGrad.join(mem)`: the layers that get `holder`
    (LayerCfg.mem_grad) accumulate their d(mem) into ONE buffer (the row-owning backward kernel writes it in the first layer
    of the backward and adds to it in the others: every sample's rows belong to one workgroup) and return no gradient for
    `mem`; this node, which the engine runs once all of them are done, hands the buffer on."""

    @staticmethod
    def forward(ctx, mem, holder):
        ctx.holder = holder
        ctx.in_dtype = mem.dtype
        ctx.set_materialize_grads(False)
        return mem.view_as(mem)

    @staticmethod
    def backward(ctx, g):
        buf = ctx.holder.pop("buf", None)
        if buf is None:
            return g, None
        if g is not None:
            buf = buf + g.float()
        if ctx.holder.get("side") and buf.dtype != ctx.in_dtype:
            # the producer (LinearLP) picks the fp32 buffer up itself: returning it here would round it to the memory's 16-bit dtype
            ctx.holder["dy32"] = buf
            return None, None
        return buf, None

    @staticmethod
    def join(mem, holder=None):
        """holder: the dict also given to the producing `LinearLP` (then flagged "side": the gradient bypasses the 16-bit edge)"""
        if holder is None:
            holder = {}
        else:
            holder["side"] = True
        return SharedMemoryGrad.apply(mem, holder), holder


class DecoderLayerFn(torch.autograd.Function):
    """One post-norm DETR decoder layer -- self-attention, norm, cross-attention, norm, FFN(ReLU), norm (detrex
    `BaseTransformerLayer` as configured at heads/tgqs_kd_detr_head/transformer.py:93-131), optionally followed by the
    decoder's shared post-norm of the layer output (transformer.py:176-183) -- as ONE autograd node over the fused kernels of
    csrc/decoder.hip (round 5): THREE launches forward (the attention block by a workgroup per sample; the FFN split over its
    hidden units; the slices' sum + norms) and FOUR backward (FFN slices incl. their weight gradients + the FFN's row sums,
    the attention block's row-owning backward, its 12 parameter gradients in one launch) instead of ~20 / ~25.
    The cross-attention contracts the source rows directly (no K / V projection, see decoder.hip); `src` is the source matrix
    (LayerCfg).  Returns (layer output, post-normed layer output | None)."""

    @staticmethod
    def forward(ctx, tgt, qpos, src, Ws, bs, Wso, bso, g0, b0, Wc, bc, Wco, bco, g1, b1n, W1, b1, W2, b2, g2, b2n, gP, bP, cfg):
        dev = qpos.device
        tgt, qpos = (None if tgt is None else tgt.contiguous()), qpos.contiguous()       # tgt None: zeros (a decoder's first layer)
        B, H, nq = cfg.B, cfg.H, cfg.nq
        train = cfg.training
        M, Fd = qpos.shape[0], W1.shape[0]
        mem = cfg.kind == "mem"
        kv_rows, kv_off = (cfg.Nv, 1) if mem else (cfg.Lk, 0)
        kpos = cfg.pos.reshape(-1, qpos.shape[1]) if cfg.pos is not None else None
        dm0 = cfg.mask_fn((B, H, nq, nq), dev) if (train and cfg.p_attn > 0) else None
        dm1 = cfg.mask_fn((B, H, nq, cfg.Lk), dev) if (train and cfg.p_attn > 0) else None
        W = [w.contiguous() for w in (Ws, bs, Wso, bso, g0, b0, Wc, bc, Wco, bco, g1, b1n)]
        sa = ops.dec_attn_fwd(tgt, qpos, W, src, B, nq, cfg.Lk, kv_rows=kv_rows, kv_off=kv_off, kpos=kpos, kpm=cfg.kpm, dm0=dm0, dm1=dm1)
        m1 = m2 = None
        if train and cfg.p_ffn > 0:
            m1, m2 = cfg.mask_fn((M, Fd), dev, cfg.p_ffn), cfg.mask_fn((M, qpos.shape[1]), dev, cfg.p_ffn)
        sf = ops.dec_ffn_fwd(sa["t2"], W1, b1, W2, b2, g2, b2n, gP=gP, bP=bP, m1=m1, m2=m2)
        # the node keeps ALIASES of its two outputs (same storage, other tensor objects): the returned tensors carry this node as
        # their grad_fn, and a node that holds them would be a reference cycle -- the carved activation buffers would live until
        # Python's cyclic collector runs instead of until the backward has used them
        ctx.cfg, ctx.sa = cfg, sa
        ctx.sf = {k: (v.detach() if (k in ("t3", "hs") and v is not None) else v) for k, v in sf.items()}
        ctx.geo = (kv_rows, kv_off)
        ctx.save_for_backward(tgt, qpos, src, kpos, dm0, dm1, m1, m2, *W, W1, W2, g2, gP)     # (None entries are allowed)
        ctx.set_materialize_grads(False)
        return sf["t3"], sf["hs"]

    @staticmethod
    def backward(ctx, d_t3, d_hs):
        tgt, qpos, src, kpos, dm0, dm1, m1, m2, *rest = ctx.saved_tensors
        W, (W1, W2, g2, gP) = rest[:12], rest[12:]
        cfg, sa, sf = ctx.cfg, ctx.sa, ctx.sf
        kv_rows, kv_off = ctx.geo
        B, nq = cfg.B, cfg.nq
        M, E = qpos.shape
        if d_t3 is None and d_hs is None:
            d_t3 = torch.zeros(M, E, device=qpos.device, dtype=torch.float32)
        d_t3 = None if d_t3 is None else d_t3.contiguous()
        d_hs = None if (d_hs is None or gP is None) else d_hs.contiguous()
        d_r3, slabs, gf = ops.dec_ffn_bwd(sf, sa["t2"], W1, W2, g2, gP=gP, d_t3=d_t3, d_hs=d_hs, m1=m1, m2=m2)
        dsrc, acc, ret_dsrc = None, False, None
        if ctx.needs_input_grad[2]:
            holder = cfg.mem_grad
            if holder is None:
                dsrc = ret_dsrc = torch.empty(src.shape[0], E, device=qpos.device, dtype=torch.float32)
            else:
                dsrc = holder.get("buf")
                acc = dsrc is not None                      # the first layer of the backward writes the buffer, the others add
                if dsrc is None:
                    dsrc = holder["buf"] = torch.empty(src.shape[0], E, device=qpos.device, dtype=torch.float32)
        d_tgt, d_qpos, ga = ops.dec_attn_bwd(sa, tgt, qpos, W, src, B, nq, cfg.Lk, dt2=d_r3, dt2_slabs=slabs, kv_rows=kv_rows, kv_off=kv_off,
                                             kpos=kpos, dm0=dm0, dm1=dm1, dsrc=dsrc, dsrc_accumulate=acc)
        dW1, db1, dW2, db2, dg2, db2n, dgP, dbP = gf
        return (d_tgt if tgt is not None else None, d_qpos, ret_dsrc, *ga, dW1, db1, dW2, db2, dg2, db2n, dgP, dbP, None)


class DecoderLayerUnfusedFn(torch.autograd.Function):
    """The same layer on the per-stage kernels of rounds 1-4 (csrc/head.hip), kept for cross-attentions with more keys than the
    fused kernels hold in LDS (`hip_ops.dec_attn_max_keys()`: e.g. patch 16 at 640 px = 1600 keys; no reference config).
    One post-norm DETR decoder layer -- self-attention, norm, cross-attention, norm, FFN(ReLU), norm (detrex
    `BaseTransformerLayer` as configured at heads/tgqs_kd_detr_head/transformer.py:93-131), optionally followed by the
    decoder's shared post-norm of the layer output (transformer.py:176-183) -- as ONE autograd node with a hand-sequenced
    backward.  The head is a chain of ~5 us launches on [B*nq, 256] rows, so its cost is the NUMBER of launches:
      * residual adds ride in GEMM epilogues (`addend`), independent GEMMs (dgrad / wgrad / bias gradient of a Linear,
        q|k and v projections ...) share one grouped launch, fan-in of gradients is folded into GEMMs
        (d_tgt = dqkv W_in + d_r1 is one K = 3E contraction; d_qpos = dqk W_qk + dxq);
      * parameter gradients are written whole; the only fill is one buffer for the LayerNorm (and, with bf16 memory,
        cross-attention K|V) gradients that are accumulated with atomics.
    ~20 launches forward and ~25 backward instead of ~26 / ~50 through per-op autograd nodes.
    kind "text": keys / values from fp32 rows xk, xv [B*Lk, E] (the TGQG layer); kind "mem": from the image memory
    mem [B*Nv, E] (16-bit -> MFMA GEMMs with cfg.wb / cfg.wbT, fp32 -> exact), key_pos cfg.pos on the patch rows.
    Returns (layer output, post-normed layer output | None)."""

    @staticmethod
    def forward(ctx, tgt, qpos, xk, xv, mem, Ws, bs, Wso, bso, g0, b0, Wc, bc, Wco, bco, g1, b1n, W1, b1, W2, b2, g2, b2n,
                gP, bP, cfg):
        dev = tgt.device
        tgt, qpos = tgt.contiguous(), qpos.contiguous()
        M, E = tgt.shape
        Fd = W1.shape[0]
        B, H, nq = cfg.B, cfg.H, cfg.nq
        train = cfg.training

        def amask(Lk):
            return cfg.mask_fn((B, H, nq, Lk), dev) if (train and cfg.p_attn > 0) else None

        def ln(x, g, b):
            _, y, mean, rstd = ops.ln_fwd(x, g, b, eps=1e-5, out_lp=False, out_f32=True)
            return y, mean, rstd

        # ---- self-attention: q = k = (tgt + qpos) W_qk (the sum is formed on the GEMM's operand load), v = tgt W_v
        qkv = _f32(M, 3 * E, device=dev)
        ops.gemm_f32_group([ops.gp(tgt, E, 1, Ws, 1, E, qkv, M, 2 * E, E, bias=bs, A2=qpos),
                            ops.gp(tgt, E, 1, Ws[2 * E:], 1, E, qkv[:, 2 * E:], M, E, E, bias=bs[2 * E:])])
        dm0 = amask(nq)
        o, P0 = ops.attn_small_fwd(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], B, H, nq, nq, kpm=None, drop=dm0, kv_rows=0)
        r1 = _f32(M, E, device=dev)
        ops.gemm_f32(o, E, 1, Wso, 1, E, r1, M, E, E, bias=bso, addend=tgt, addend_rows=M)         # tgt + out_proj(o)
        t1, mean1, rstd1 = ln(r1, g0, b0)
        # ---- cross-attention: q = (t1 + qpos) W_q
        q = _f32(M, E, device=dev)
        if cfg.kind == "text":
            xk, xv = xk.contiguous(), xv.contiguous()
            R, Lk = xk.shape[0], cfg.Lk
            kv = _f32(R, 2 * E, device=dev)
            ops.gemm_f32_group([ops.gp(t1, E, 1, Wc, 1, E, q, M, E, E, bias=bc, A2=qpos),
                                ops.gp(xk, E, 1, Wc[E:], 1, E, kv, R, E, E, bias=bc[E:]),
                                ops.gp(xv, E, 1, Wc[2 * E:], 1, E, kv[:, E:], R, E, E, bias=bc[2 * E:])])
            dm1 = amask(Lk)
            o2, P1 = ops.attn_small_fwd(q, kv[:, :E], kv[:, E:], B, H, nq, Lk, kpm=cfg.kpm, drop=dm1, kv_rows=0)
            pos2 = posk = None
        else:
            Nv = cfg.Nv
            HW, R = Nv - 1, B * Nv
            pos2 = cfg.pos.reshape(-1, E)
            posk = _f32(pos2.shape[0], E, device=dev)
            probs = [ops.gp(t1, E, 1, Wc, 1, E, q, M, E, E, bias=bc, A2=qpos),
                     ops.gp(pos2, E, 1, Wc[E:], 1, E, posk, pos2.shape[0], E, E)]
            if mem.dtype == ops.LP():
                kv = ops.gemm_nt(mem, cfg.wb, bias=bc[E:], out_dtype=torch.float32)                  # [B*Nv, 2E]
            else:
                kv = _f32(R, 2 * E, device=dev)
                probs.append(ops.gp(mem, mem.stride(0), 1, Wc[E:], 1, E, kv, R, 2 * E, E, bias=bc[E:]))
            ops.gemm_f32_group(probs)
            dm1 = amask(HW)
            # keys of sample b start at row b*Nv + 1: views that begin at row 1, batch stride Nv rows; key_pos (patch keys
            # only) joins the projected keys on the kernel's K load
            o2, P1 = ops.attn_small_fwd(q, kv[1:, :E], kv[1:, E:], B, H, nq, HW, kpm=cfg.kpm, drop=dm1, kv_rows=Nv, kpos=posk)
        r2 = _f32(M, E, device=dev)
        ops.gemm_f32(o2, E, 1, Wco, 1, E, r2, M, E, E, bias=bco, addend=t1, addend_rows=M)
        t2, mean2, rstd2 = ln(r2, g1, b1n)
        # ---- FFN: Linear, ReLU, dropout, Linear, dropout, + identity; the dropout multipliers ride in the GEMM epilogues
        h1d, r3 = _f32(M, Fd, device=dev), _f32(M, E, device=dev)
        if train and cfg.p_ffn > 0:
            m1, m2 = cfg.mask_fn((M, Fd), dev, cfg.p_ffn), cfg.mask_fn((M, E), dev, cfg.p_ffn)
            ops.gemm_f32_group([ops.gp(t2, E, 1, W1, 1, E, h1d, M, Fd, E, bias=b1, act=2, mult=m1)])
            ops.gemm_f32_group([ops.gp(h1d, Fd, 1, W2, 1, Fd, r3, M, E, Fd, bias=b2, mult=m2, addend=t2, addend_rows=M)])
        else:
            m1 = m2 = None
            ops.gemm_f32(t2, E, 1, W1, 1, E, h1d, M, Fd, E, bias=b1, act=2)
            ops.gemm_f32(h1d, Fd, 1, W2, 1, Fd, r3, M, E, Fd, bias=b2, addend=t2, addend_rows=M)
        t3, mean3, rstd3 = ln(r3, g2, b2n)
        hs = meanP = rstdP = None
        if gP is not None:
            hs, meanP, rstdP = ln(t3, gP, bP)
        ctx.save_for_backward(tgt, qpos, qkv, P0, dm0, o, r1, mean1, rstd1, t1, q, kv, P1, dm1, o2, r2, mean2, rstd2, t2,
                              h1d, m1, m2, r3, mean3, rstd3, t3, meanP, rstdP, xk, xv, mem, pos2, posk,
                              Ws, Wso, g0, Wc, Wco, g1, W1, W2, g2, gP)
        ctx.cfg = cfg
        ctx.set_materialize_grads(False)
        return t3, hs

    @staticmethod
    def backward(ctx, d_t3, d_hs):
        (tgt, qpos, qkv, P0, dm0, o, r1, mean1, rstd1, t1, q, kv, P1, dm1, o2, r2, mean2, rstd2, t2, h1d, m1, m2, r3,
         mean3, rstd3, t3, meanP, rstdP, xk, xv, mem, pos2, posk, Ws, Wso, g0, Wc, Wco, g1, W1, W2, g2,
         gP) = ctx.saved_tensors
        cfg = ctx.cfg
        dev = tgt.device
        M, E = tgt.shape
        Fd = W1.shape[0]
        B, H, nq = cfg.B, cfg.H, cfg.nq
        need = ctx.needs_input_grad
        ones = _ones(M, dev)
        mem_bf = cfg.kind == "mem" and mem.dtype == ops.LP()
        # the one fill of this backward: LayerNorm gradients (atomics) and, with bf16 memory, the cross-attention in_proj
        # gradient that the bf16 wgrad kernel accumulates into
        z = torch.zeros(8 * E + (3 * E * E + 3 * E if mem_bf else 0), device=dev, dtype=torch.float32)
        dg2, db2n, dg1, db1n, dg0, db0, dgP, dbP = (z[i * E:(i + 1) * E] for i in range(8))
        if d_t3 is not None:
            d_t3 = d_t3.contiguous()
        if gP is not None and d_hs is not None:
            dx = _f32(M, E, device=dev)
            ops.ln_bwd(d_hs.contiguous(), t3, meanP, rstdP, gP, dgP, dbP, dres=d_t3, dx_f32=dx)
            d_t3 = dx
        else:
            dgP = dbP = None
        if d_t3 is None:
            d_t3 = torch.zeros(M, E, device=dev, dtype=torch.float32)
        # ---- FFN
        d_r3 = _f32(M, E, device=dev)
        ops.ln_bwd(d_t3, r3, mean3, rstd3, g2, dg2, db2n, dx_f32=d_r3)
        d_h2 = d_r3 * m2 if m2 is not None else d_r3
        d_h1 = _f32(M, Fd, device=dev)
        dW2, db2 = _f32(E, Fd, device=dev), _f32(1, E, device=dev)
        # d(pre-activation) = (d_h2 W2) * dropout multiplier where the ReLU was open: h1d = relu(.) * m1 > 0 <=> both
        ops.gemm_f32_group([ops.gp(d_h2, E, 1, W2, Fd, 1, d_h1, M, Fd, E, gate=h1d, mult=m1),
                            ops.gp(d_h2, 1, E, h1d, Fd, 1, dW2, E, Fd, M),
                            ops.gp(ones, 0, 1, d_h2, E, 1, db2, 1, E, M)])
        d_t2 = _f32(M, E, device=dev)
        dW1, db1 = _f32(Fd, E, device=dev), _f32(1, Fd, device=dev)
        ops.gemm_f32_group([ops.gp(d_h1, Fd, 1, W1, E, 1, d_t2, M, E, Fd, addend=d_r3, addend_rows=M),
                            ops.gp(d_h1, 1, Fd, t2, E, 1, dW1, Fd, E, M),
                            ops.gp(ones, 0, 1, d_h1, Fd, 1, db1, 1, Fd, M)])
        # ---- cross-attention
        d_r2 = _f32(M, E, device=dev)
        ops.ln_bwd(d_t2, r2, mean2, rstd2, g1, dg1, db1n, dx_f32=d_r2)
        d_o2 = _f32(M, E, device=dev)
        dWco, dbco = _f32(E, E, device=dev), _f32(1, E, device=dev)
        ops.gemm_f32_group([ops.gp(d_r2, E, 1, Wco, E, 1, d_o2, M, E, E),
                            ops.gp(d_r2, 1, E, o2, E, 1, dWco, E, E, M),
                            ops.gp(ones, 0, 1, d_r2, E, 1, dbco, 1, E, M)])
        dq, dxq, d_t1 = _f32(M, E, device=dev), _f32(M, E, device=dev), _f32(M, E, device=dev)
        dxk = dxv = dmem = None
        if cfg.kind == "text":
            R, Lk = xk.shape[0], cfg.Lk
            dkv = _f32(R, 2 * E, device=dev)
            ops.attn_small_bwd(q, kv[:, :E], kv[:, E:], P1, d_o2, dq, dkv[:, :E], dkv[:, E:], B, H, nq, Lk, kpm=cfg.kpm,
                               drop=dm1, kv_rows=0)
            dWc, dbc = _f32(3 * E, E, device=dev), _f32(1, 3 * E, device=dev)
            probs = [ops.gp(dq, E, 1, Wc, E, 1, dxq, M, E, E),
                     ops.gp(dq, E, 1, Wc, E, 1, d_t1, M, E, E, addend=d_r2, addend_rows=M),
                     ops.gp(dq, 1, E, t1, E, 1, dWc, E, E, M, B2=qpos),
                     ops.gp(dkv, 1, 2 * E, xk, E, 1, dWc[E:], E, E, R),
                     ops.gp(dkv[:, E:], 1, 2 * E, xv, E, 1, dWc[2 * E:], E, E, R),
                     ops.gp(ones, 0, 1, dq, E, 1, dbc, 1, E, M),
                     ops.gp(_ones(R, dev), 0, 1, dkv, 2 * E, 1, dbc[:, E:], 1, 2 * E, R)]
            if need[2]:
                dxk = _f32(R, E, device=dev)
                probs.append(ops.gp(dkv, 2 * E, 1, Wc[E:], E, 1, dxk, R, E, E))
            if need[3]:
                dxv = _f32(R, E, device=dev)
                probs.append(ops.gp(dkv[:, E:], 2 * E, 1, Wc[2 * E:], E, 1, dxv, R, E, E))
            ops.gemm_f32_group(probs)
        else:
            Nv = cfg.Nv
            HW, R = Nv - 1, B * Nv
            dkv = _f32(R, 2 * E, device=dev)
            dkv.view(B, Nv, 2 * E)[:, 0].zero_()                                         # CLS rows: not keys, no gradient
            ops.attn_small_bwd(q, kv[1:, :E], kv[1:, E:], P1, d_o2, dq, dkv[1:, :E], dkv[1:, E:], B, H, nq, HW, kpm=cfg.kpm,
                               drop=dm1, kv_rows=Nv, kpos=posk)
            if mem_bf:
                dWc, dbc = z[8 * E:8 * E + 3 * E * E].view(3 * E, E), z[8 * E + 3 * E * E:].view(1, 3 * E)
            else:
                dWc, dbc = _f32(3 * E, E, device=dev), _f32(1, 3 * E, device=dev)
            probs = [ops.gp(dq, E, 1, Wc, E, 1, dxq, M, E, E),
                     ops.gp(dq, E, 1, Wc, E, 1, d_t1, M, E, E, addend=d_r2, addend_rows=M),
                     ops.gp(dq, 1, E, t1, E, 1, dWc, E, E, M, B2=qpos),
                     ops.gp(ones, 0, 1, dq, E, 1, dbc, 1, E, M)]
            if mem_bf:
                ops.gemm_f32_group(probs)
                S = ops.grad_scale()                                                     # 16-bit operand dkv * S
                dkvb = ops.cast_lp(dkv, scale=S)
                if need[4]:
                    acc = cfg.mem_grad
                    if acc is None:
                        dmem = ops.gemm_nt(dkvb, cfg.wbT, out_dtype=torch.float32, alpha=1.0 / S)   # [R, E] fp32, true gradient
                    elif acc.get("buf") is None:
                        acc["buf"] = ops.gemm_nt(dkvb, cfg.wbT, out_dtype=torch.float32, alpha=1.0 / S)
                    else:       # a later layer's backward already left its share: buf += dkv W (residual epilogue, in place)
                        ops.gemm_nt(dkvb, cfg.wbT, out=acc["buf"], residual=acc["buf"], alpha=1.0 / S)
                ops.gemm_tn(dkvb, mem, dWc[E:], db=dbc[0, E:], out_scale=1.0 / S)
            else:
                if need[4]:
                    dmem = _f32(R, E, device=dev)
                    probs.append(ops.gp(dkv, 2 * E, 1, Wc[E:], E, 1, dmem, R, E, 2 * E))
                probs.append(ops.gp(dkv, 1, 2 * E, mem, mem.stride(0), 1, dWc[E:], 2 * E, E, R))
                probs.append(ops.gp(_ones(R, dev), 0, 1, dkv, 2 * E, 1, dbc[:, E:], 1, 2 * E, R))
                ops.gemm_f32_group(probs)
            dk3 = dkv.view(B, Nv, 2 * E)[:, 1:, :E]
            dpk = dk3.sum(0) if cfg.pos.dim() == 2 else dk3.reshape(-1, E)               # d(pos Wk^T)
            ops.gemm_f32(dpk, 1, E, pos2, E, 1, dWc[E:2 * E], E, E, pos2.shape[0], accumulate=True)
        # ---- self-attention
        d_r1 = _f32(M, E, device=dev)
        ops.ln_bwd(d_t1, r1, mean1, rstd1, g0, dg0, db0, dx_f32=d_r1)
        d_o = _f32(M, E, device=dev)
        dWso, dbso = _f32(E, E, device=dev), _f32(1, E, device=dev)
        ops.gemm_f32_group([ops.gp(d_r1, E, 1, Wso, E, 1, d_o, M, E, E),
                            ops.gp(d_r1, 1, E, o, E, 1, dWso, E, E, M),
                            ops.gp(ones, 0, 1, d_r1, E, 1, dbso, 1, E, M)])
        dqkv = _f32(M, 3 * E, device=dev)
        ops.attn_small_bwd(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], P0, d_o, dqkv[:, :E], dqkv[:, E:2 * E], dqkv[:, 2 * E:],
                           B, H, nq, nq, kpm=None, drop=dm0, kv_rows=0)
        d_qpos = _f32(M, E, device=dev)
        dWs, dbs = _f32(3 * E, E, device=dev), _f32(1, 3 * E, device=dev)
        probs = [ops.gp(dqkv, 3 * E, 1, Ws, E, 1, d_qpos, M, E, 2 * E, addend=dxq, addend_rows=M),   # dqk W_qk + dxq
                 ops.gp(dqkv, 1, 3 * E, tgt, E, 1, dWs, 2 * E, E, M, B2=qpos),
                 ops.gp(dqkv[:, 2 * E:], 1, 3 * E, tgt, E, 1, dWs[2 * E:], E, E, M),
                 ops.gp(ones, 0, 1, dqkv, 3 * E, 1, dbs, 1, 3 * E, M)]
        d_tgt = None
        if need[0]:
            d_tgt = _f32(M, E, device=dev)                                               # dqkv W_in (K = 3E) + d_r1
            probs.append(ops.gp(dqkv, 3 * E, 1, Ws, E, 1, d_tgt, M, E, 3 * E, addend=d_r1, addend_rows=M))
        ops.gemm_f32_group(probs)
        return (d_tgt, d_qpos, dxk, dxv, dmem, dWs, dbs.view(3 * E), dWso, dbso.view(E), dg0, db0, dWc, dbc.view(3 * E), dWco,
                dbco.view(E), dg1, db1n, dW1, db1.view(Fd), dW2, db2.view(E), dg2, db2n, dgP, dbP, None)


class Criterion(torch.autograd.Function):
    """SetCriterion value + analytic gradients in one launch; `coef_mode`/`coef` carry the branch weight."""

    @staticmethod
    def forward(ctx, logits, boxes, match_idx, tboxes, tlabels, num_boxes, wdist, coef_mode, coef, eos, weights):
        out, dl, db = ops.criterion(logits.contiguous(), boxes.contiguous(), match_idx, tboxes, tlabels, num_boxes,
                                    wdist, coef_mode, coef, eos, weights)
        ctx.save_for_backward(dl, db)
        ctx.mark_non_differentiable(out[1:])
        return out[0], out[1:]

    @staticmethod
    def backward(ctx, g, _g2):
        dl, db = ctx.saved_tensors
        if dl.untyped_storage().data_ptr() == db.untyped_storage().data_ptr() and dl.storage_offset() == 0 \
                and db.storage_offset() == dl.numel() and dl.is_contiguous() and db.is_contiguous():
            both = g * dl.new_empty(0).set_(dl.untyped_storage(), 0, (dl.numel() + db.numel(),))     # [dlogits | dboxes]: one launch
            return (both[:dl.numel()].view(dl.shape), both[dl.numel():].view(db.shape)) + (None,) * 9
        return g * dl, g * db, None, None, None, None, None, None, None, None, None


class SplitEncoderOutput(torch.autograd.Function):
    """Encoder output (modality-major rows) as fp32 `out32` [B*Nv + B*T, D] and its 16-bit copy `out_lp` -> (vision
    rows [B*Nv, D] 16-bit view = the A operand of the memory projection, text fp32 [B*T, D], cls fp32 [B, D]); the
    backward assembles the single fp32 gradient the encoder engine expects."""

    @staticmethod
    def forward(ctx, out32, out_lp, B, Nv, T, xholder=None):
        D = out32.shape[1]
        ctx.geo = (B, Nv, T, D)
        ctx.xholder = xholder            # where the consumer of `vis` (LinearLP) leaves the fp32 gradient of these 16-bit rows
        ctx.set_materialize_grads(False)
        vis = out_lp[:B * Nv]
        text = out32[B * Nv:].clone()
        cls = out32[:B * Nv].view(B, Nv, D)[:, 0].clone()
        return vis, text, cls

    @staticmethod
    def backward(ctx, dvis, dtext, dcls):
        B, Nv, T, D = ctx.geo
        side = ctx.xholder.pop("dx32", None) if ctx.xholder is not None else None
        if side is not None:
            dvis = side if dvis is None else side + dvis.float()
        ref = dvis if dvis is not None else (dtext if dtext is not None else dcls)
        if ref is None:
            return None, None, None, None, None, None
        rows = B * (Nv + T)
        if dvis is not None and dvis.dtype == torch.float32 and dvis.is_contiguous() and dvis.storage_offset() == 0 \
                and tuple(dvis.shape) == (B * Nv, D) and dvis.untyped_storage().nbytes() >= rows * D * 4:
            d = dvis.new_empty(0).set_(dvis.untyped_storage(), 0, (rows, D))      # LinearLP.backward left room for the text rows
        else:
            d = torch.empty(rows, D, device=ref.device, dtype=torch.float32)
            if dvis is not None:
                d[:B * Nv] = dvis
            else:
                d[:B * Nv].zero_()
        if dtext is not None:
            d[B * Nv:] = dtext
        else:
            d[B * Nv:].zero_()
        if dcls is not None:
            d[:B * Nv].view(B, Nv, D)[:, 0] += dcls
        return d, None, None, None, None, None
