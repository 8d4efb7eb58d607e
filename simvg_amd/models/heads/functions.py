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
            dy = dy * (y > 0)
        M, K = x.shape
        N = W.shape[0]
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, device=dy.device, dtype=torch.float32)
            ops.gemm_f32(dy, N, 1, W, W.stride(0), 1, dx, M, K, N)                      # dx = dy W
        if ctx.needs_input_grad[1]:
            dW = torch.empty(N, K, device=dy.device, dtype=torch.float32)
            ops.gemm_f32(dy, 1, N, x, x.stride(0), x.stride(1), dW, N, K, M)            # dW = dy^T x
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = torch.empty(1, N, device=dy.device, dtype=torch.float32)
            ops.gemm_f32(_ones(M, dy.device), 0, 1, dy, N, 1, db, 1, N, M)              # db = 1^T dy
            db = db.view(N)
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
        dg, db = torch.zeros_like(g), torch.zeros_like(g)
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
