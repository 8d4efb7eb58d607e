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
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, y = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.relu:
            dy = torch.ops.aten.threshold_backward(dy, y, 0.0)
        M, K = x.shape
        N = W.shape[0]
        dx = dW = db = None
        if ctx.needs_input_grad[1]:
            dW = torch.empty(N, K, device=dy.device, dtype=torch.float32)
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = torch.empty(1, N, device=dy.device, dtype=torch.float32)
        if dW is not None:
            ops.gemm_f32(dy, 1, N, x, x.stride(0), x.stride(1), dW, N, K, M)        # dW = dy^T x
        if db is not None:
            ops.gemm_f32(_ones(M, dy.device), 0, 1, dy, N, 1, db, 1, N, M)          # db = 1^T dy
            db = db.view(N)
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, device=dy.device, dtype=torch.float32)
            ops.gemm_f32(dy, N, 1, W, W.stride(0), 1, dx, M, K, N)                      # dx = dy W
        return dx, dW, db, None


class LinearBF16(torch.autograd.Function):
    """y = x W^T + b on the bf16 MFMA GEMM for the B*(1+HW) memory rows.  x bf16 [M,K]; W fp32 master
    [N,K]; the bf16 copies (plain and transposed) are refreshed by the caller's weight-prep."""

    @staticmethod
    def forward(ctx, x, W, b, w_bf16, wT_bf16, out_bf16):
        y = ops.gemm_nt(x, w_bf16, bias=b, out_dtype=torch.bfloat16 if out_bf16 else torch.float32)
        ctx.save_for_backward(x, wT_bf16)
        ctx.shapeW = W.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wT = ctx.saved_tensors
        dyb = dy.contiguous() if dy.dtype == torch.bfloat16 else ops.cast_bf16(dy.contiguous())
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(dyb, wT)                                                    # [M,K] bf16
        if ctx.needs_input_grad[1]:
            dW = torch.zeros(ctx.shapeW, device=dy.device, dtype=torch.float32)
            db = torch.zeros(ctx.shapeW[0], device=dy.device, dtype=torch.float32) if ctx.needs_input_grad[2] else None
            ops.gemm_tn(dyb, x, dW.view(ctx.shapeW[0], -1), db=db)
        elif ctx.needs_input_grad[2]:
            db = torch.zeros(ctx.shapeW[0], device=dy.device, dtype=torch.float32)
            ops.colsum(dyb, db)
        return dx, dW, db, None, None, None


class LayerNormF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, eps):
        x = x.contiguous()
        _, y, mean, rstd = ops.ln_fwd(x, g, b, eps=eps, out_bf16=False, out_f32=True)
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


class SelfAttnBlock(torch.autograd.Function):
    """in_proj + attention core of an nn.MultiheadAttention self-attention (detrex `MultiheadAttention`, query = key =
    x + query_pos, value = x): q|k = x_qk W[:2E]^T + b[:2E], v = x_v W[2E:]^T + b[2E:], softmax(q k^T / sqrt(32)) v per
    (sample, head).  ONE autograd node: the gradient of the packed in_proj_weight / in_proj_bias is produced whole
    (row blocks written by their own GEMMs) instead of being assembled from sliced views by ~15 fill / copy / add
    launches."""

    @staticmethod
    def forward(ctx, x_qk, x_v, W, b, B, H, L, drop):
        x_qk, x_v = x_qk.contiguous(), x_v.contiguous()
        M, E = x_v.shape
        qkv = _f32(M, 3 * E, device=x_v.device)
        ops.gemm_f32(x_qk, E, 1, W, 1, E, qkv, M, 2 * E, E, bias=b)
        ops.gemm_f32(x_v, E, 1, W[2 * E:], 1, E, qkv[:, 2 * E:], M, E, E, bias=b[2 * E:])
        out, P = ops.attn_small_fwd(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], B, H, L, L, kpm=None, drop=drop, kv_rows=0)
        ctx.save_for_backward(x_qk, x_v, W, qkv, P, drop)
        ctx.geo = (B, H, L)
        return out

    @staticmethod
    def backward(ctx, dout):
        x_qk, x_v, W, qkv, P, drop = ctx.saved_tensors
        B, H, L = ctx.geo
        M, E = x_v.shape
        dev = dout.device
        dqkv = _f32(M, 3 * E, device=dev)
        ops.attn_small_bwd(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], P, dout.contiguous(), dqkv[:, :E], dqkv[:, E:2 * E],
                           dqkv[:, 2 * E:], B, H, L, L, kpm=None, drop=drop, kv_rows=0)
        dW, db = _f32(3 * E, E, device=dev), _f32(1, 3 * E, device=dev)
        ops.gemm_f32(dqkv, 1, 3 * E, x_qk, E, 1, dW, 2 * E, E, M)                   # dW[:2E] = dqk^T x_qk
        ops.gemm_f32(dqkv[:, 2 * E:], 1, 3 * E, x_v, E, 1, dW[2 * E:], E, E, M)     # dW[2E:] = dv^T x_v
        ops.gemm_f32(_ones(M, dev), 0, 1, dqkv, 3 * E, 1, db, 1, 3 * E, M)          # db = 1^T dqkv
        dx_qk = dx_v = None
        if ctx.needs_input_grad[0]:
            dx_qk = _f32(M, E, device=dev)
            ops.gemm_f32(dqkv, 3 * E, 1, W, E, 1, dx_qk, M, E, 2 * E)
        if ctx.needs_input_grad[1]:
            dx_v = _f32(M, E, device=dev)
            ops.gemm_f32(dqkv[:, 2 * E:], 3 * E, 1, W[2 * E:], E, 1, dx_v, M, E, E)
        return dx_qk, dx_v, dW, db.view(3 * E), None, None, None, None


class CrossAttnBlock(torch.autograd.Function):
    """in_proj + core of a cross-attention whose keys / values are fp32 rows (the TGQG layer's text cross-attention):
    q = xq W[:E]^T + b[:E] on B*Lq rows, k = xk W[E:2E]^T + b[E:2E], v = xv W[2E:]^T + b[2E:] on B*Lk rows."""

    @staticmethod
    def forward(ctx, xq, xk, xv, W, b, B, H, Lq, Lk, kpm, drop):
        xq, xk, xv = xq.contiguous(), xk.contiguous(), xv.contiguous()
        M, E = xq.shape
        R = xk.shape[0]
        q, kv = _f32(M, E, device=xq.device), _f32(R, 2 * E, device=xq.device)
        ops.gemm_f32(xq, E, 1, W, 1, E, q, M, E, E, bias=b)
        ops.gemm_f32(xk, E, 1, W[E:], 1, E, kv, R, E, E, bias=b[E:])
        ops.gemm_f32(xv, E, 1, W[2 * E:], 1, E, kv[:, E:], R, E, E, bias=b[2 * E:])
        out, P = ops.attn_small_fwd(q, kv[:, :E], kv[:, E:], B, H, Lq, Lk, kpm=kpm, drop=drop, kv_rows=0)
        ctx.save_for_backward(xq, xk, xv, W, q, kv, P, kpm, drop)
        ctx.geo = (B, H, Lq, Lk)
        return out

    @staticmethod
    def backward(ctx, dout):
        xq, xk, xv, W, q, kv, P, kpm, drop = ctx.saved_tensors
        B, H, Lq, Lk = ctx.geo
        M, E = xq.shape
        R = xk.shape[0]
        dev = dout.device
        dq, dkv = _f32(M, E, device=dev), _f32(R, 2 * E, device=dev)
        ops.attn_small_bwd(q, kv[:, :E], kv[:, E:], P, dout.contiguous(), dq, dkv[:, :E], dkv[:, E:], B, H, Lq, Lk,
                           kpm=kpm, drop=drop, kv_rows=0)
        dW, db = _f32(3 * E, E, device=dev), _f32(1, 3 * E, device=dev)
        ops.gemm_f32(dq, 1, E, xq, E, 1, dW, E, E, M)
        ops.gemm_f32(dkv, 1, 2 * E, xk, E, 1, dW[E:], E, E, R)
        ops.gemm_f32(dkv[:, E:], 1, 2 * E, xv, E, 1, dW[2 * E:], E, E, R)
        ops.gemm_f32(_ones(M, dev), 0, 1, dq, E, 1, db, 1, E, M)
        ops.gemm_f32(_ones(R, dev), 0, 1, dkv, 2 * E, 1, db[:, E:], 1, 2 * E, R)
        dxq = dxk = dxv = None
        if ctx.needs_input_grad[0]:
            dxq = _f32(M, E, device=dev)
            ops.gemm_f32(dq, E, 1, W, E, 1, dxq, M, E, E)
        if ctx.needs_input_grad[1]:
            dxk = _f32(R, E, device=dev)
            ops.gemm_f32(dkv, 2 * E, 1, W[E:], E, 1, dxk, R, E, E)
        if ctx.needs_input_grad[2]:
            dxv = _f32(R, E, device=dev)
            ops.gemm_f32(dkv[:, E:], 2 * E, 1, W[2 * E:], E, 1, dxv, R, E, E)
        return dxq, dxk, dxv, dW, db.view(3 * E), None, None, None, None, None, None


class MemCrossAttnBlock(torch.autograd.Function):
    """in_proj + core of a decoder layer's cross-attention over the image memory (transformer.py:134-186 via detrex
    `MultiheadAttention`: key = memory + key_pos, value = memory).  mem [B*Nv, E] holds every vision row (the CLS row of
    each sample is carried along and never used as a key); K = mem Wk^T + bk + pos Wk^T, V = mem Wv^T + bv.
    mem bf16 -> the K|V projection, its dgrad and wgrad run on the bf16 MFMA GEMMs (wb = bf16 W[E:], wbT its transpose);
    mem fp32 (precision="fp32") -> exact fp32 GEMMs.  pos: [HW, E] shared by the batch or [B, HW, E]."""

    @staticmethod
    def forward(ctx, xq, mem, W, b, pos, wb, wbT, B, H, Lq, Nv, kpm, drop):
        xq = xq.contiguous()
        M, E = xq.shape
        HW = Nv - 1
        dev = xq.device
        q = _f32(M, E, device=dev)
        ops.gemm_f32(xq, E, 1, W, 1, E, q, M, E, E, bias=b)
        if mem.dtype == torch.bfloat16:
            kv = ops.gemm_nt(mem, wb, bias=b[E:], out_dtype=torch.float32)               # [B*Nv, 2E]
        else:
            kv = _f32(B * Nv, 2 * E, device=dev)
            ops.gemm_f32(mem, mem.stride(0), 1, W[E:], 1, E, kv, B * Nv, 2 * E, E, bias=b[E:])
        pos2 = pos.reshape(-1, E)
        posk = _f32(pos2.shape[0], E, device=dev)
        ops.gemm_f32(pos2, E, 1, W[E:], 1, E, posk, pos2.shape[0], E, E)
        kv.view(B, Nv, 2 * E)[:, 1:, :E].add_(posk.view(-1, HW, E))                      # key_pos on the patch keys only
        # keys of sample b start at row b*Nv + 1: views that begin at row 1, batch stride Nv rows
        out, P = ops.attn_small_fwd(q, kv[1:, :E], kv[1:, E:], B, H, Lq, HW, kpm=kpm, drop=drop, kv_rows=Nv)
        ctx.save_for_backward(xq, mem, W, pos2, wbT, q, kv, P, kpm, drop)
        ctx.geo = (B, H, Lq, Nv, pos.dim() == 2)
        return out

    @staticmethod
    def backward(ctx, dout):
        xq, mem, W, pos2, wbT, q, kv, P, kpm, drop = ctx.saved_tensors
        B, H, Lq, Nv, shared_pos = ctx.geo
        M, E = xq.shape
        HW, R = Nv - 1, B * Nv
        dev = dout.device
        dq, dkv = _f32(M, E, device=dev), _f32(R, 2 * E, device=dev)
        dkv.view(B, Nv, 2 * E)[:, 0].zero_()                                             # CLS rows: not keys, no gradient
        ops.attn_small_bwd(q, kv[1:, :E], kv[1:, E:], P, dout.contiguous(), dq, dkv[1:, :E], dkv[1:, E:], B, H, Lq, HW,
                           kpm=kpm, drop=drop, kv_rows=Nv)
        dmem = None
        bf = mem.dtype == torch.bfloat16
        if bf:
            dW = torch.zeros(3 * E, E, device=dev, dtype=torch.float32)                  # gemm_tn accumulates
            db = torch.zeros(1, 3 * E, device=dev, dtype=torch.float32)
        else:
            dW, db = _f32(3 * E, E, device=dev), _f32(1, 3 * E, device=dev)
        # the memory-row side: rows [E:] of dW / db
        if bf:
            dkvb = ops.cast_bf16(dkv)
            if ctx.needs_input_grad[1]:
                dmem = ops.gemm_nt(dkvb, wbT)                                        # [R, E] bf16
            ops.gemm_tn(dkvb, mem, dW[E:], db=db[0, E:])
        else:
            if ctx.needs_input_grad[1]:
                dmem = _f32(R, E, device=dev)
                ops.gemm_f32(dkv, 2 * E, 1, W[E:], E, 1, dmem, R, E, 2 * E)
            ops.gemm_f32(dkv, 1, 2 * E, mem, mem.stride(0), 1, dW[E:], 2 * E, E, R)
            ops.gemm_f32(_ones(R, dev), 0, 1, dkv, 2 * E, 1, db[:, E:], 1, 2 * E, R)
        dk3 = dkv.view(B, Nv, 2 * E)[:, 1:, :E]
        dpk = dk3.sum(0) if shared_pos else dk3.reshape(-1, E)                       # d(pos Wk^T)
        ops.gemm_f32(dpk, 1, E, pos2, E, 1, dW[E:2 * E], E, E, pos2.shape[0], accumulate=True)
        dxq = None
        if ctx.needs_input_grad[0]:                                                      # the query-row side: rows [:E]
            dxq = _f32(M, E, device=dev)
            ops.gemm_f32(dq, E, 1, W, E, 1, dxq, M, E, E)
        ops.gemm_f32(dq, 1, E, xq, E, 1, dW, E, E, M)
        ops.gemm_f32(_ones(M, dev), 0, 1, dq, E, 1, db, 1, E, M)
        return dxq, dmem, dW, db.view(3 * E), None, None, None, None, None, None, None, None, None


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
        return g * dl, g * db, None, None, None, None, None, None, None, None, None


class SplitEncoderOutput(torch.autograd.Function):
    """enc_out [B*Nv + B*T, D] bf16 (modality-major) -> (vision rows bf16 [B*Nv, D] view, text fp32 [B*T, D],
    cls fp32 [B, D]); backward assembles the single bf16 gradient the encoder engine expects."""

    @staticmethod
    def forward(ctx, out, B, Nv, T):
        D = out.shape[1]
        ctx.geo = (B, Nv, T, D)
        vis = out[:B * Nv]
        text = out[B * Nv:].float()
        cls = vis.view(B, Nv, D)[:, 0].float()
        return vis, text, cls

    @staticmethod
    def backward(ctx, dvis, dtext, dcls):
        B, Nv, T, D = ctx.geo
        d = torch.empty(B * (Nv + T), D, device=dtext.device if dtext is not None else dvis.device, dtype=torch.bfloat16)
        if dvis is not None:
            d[:B * Nv] = dvis
        else:
            d[:B * Nv].zero_()
        if dtext is not None:
            d[B * Nv:] = dtext
        else:
            d[B * Nv:].zero_()
        if dcls is not None:
            d[:B * Nv].view(B, Nv, D)[:, 0] += dcls.to(torch.bfloat16)
        return d, None, None, None
