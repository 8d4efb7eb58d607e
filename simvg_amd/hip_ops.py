"""Tensor-level wrappers over the C-ABI (one Python function per exported kernel entry point).

Pointers are `tensor.data_ptr()` of caller-owned PyTorch-ROCm tensors; every call enqueues on the
current HIP stream and never synchronises.  PyTorch here is device memory + streams only.
"""
import ctypes as C

import numpy

import os

import torch

from . import _lib

_LP = None
_GRAD_SCALE = None


def LP():
    """torch dtype of the library's 16-bit operand / storage format (float16 unless built with SIMVG_LOWP_BF16)"""
    global _LP
    if _LP is None:
        _LP = torch.float16 if _lib.lowp_format() == "fp16" else torch.bfloat16
    return _LP


def grad_scale():
    """Power-of-two scale S carried by every 16-bit tensor of a backward pass (fp16 has 5 exponent bits: at B = 64 the
    activation gradients of this model sit at 1e-7 .. 1e-3, i.e. in fp16's subnormal range unscaled).  The scale is
    removed where parameter gradients are written (`out_scale` / `param_scale` / `alpha` arguments), so `.grad` tensors
    are true gradients, and because S is a power of two the results do not depend on it as long as nothing leaves fp16's
    normal range.  S follows the gradient the encoder receives (`GradScaleTracker`: the largest |d loss / d encoder
    output| of the step before last is brought to ~2^6, three orders of magnitude below the saturation value 65504), starting
    from 2^10; `set_grad_scale(v)` pins it.  bf16 builds: always 1."""
    global _GRAD_SCALE
    if _GRAD_SCALE is None:
        _GRAD_SCALE = 1024.0 if LP() == torch.float16 else 1.0
    return _GRAD_SCALE


_GRAD_SCALE_PINNED = False


def set_grad_scale(v):
    """pin the backward's 16-bit gradient scale (a power of two); None returns to the tracked scale"""
    global _GRAD_SCALE, _GRAD_SCALE_PINNED
    _GRAD_SCALE_PINNED = v is not None
    _GRAD_SCALE = None if v is None else float(v)


class GradScaleTracker:
    """Keeps S where the incoming gradient needs it, DETERMINISTICALLY: `observe(dout)` (called at the entry of the encoder
    backward of step k) queues max|dout| -> pinned host memory behind an event, into slot k % 2; `update()` (called at the
    training forward of step k + 2) waits for exactly that event and sets S = 2^floor(log2(64 / max|dout|)), clamped to
    [1, 2^24].  So the scale of step n is a function of the gradient of step n - 2 and of nothing else -- not of whether a copy
    happened to have completed (an earlier version polled the event).  The wait is on work queued two steps earlier: it blocks
    only a host that is more than a step ahead of the device, which costs no throughput.  A step whose gradient jumps by > 2^9
    before the scale follows saturates (stores clamp at +-65504, no inf).  S itself is process-wide (`grad_scale()`); with
    several training models in one process the last update wins -- results do not depend on S inside fp16's normal range."""

    TARGET = 64.0

    def __init__(self):
        self._host = None
        self._events = [None, None]
        self._seen = 0                     # observations queued so far
        self._used = 0                     # observations consumed so far

    def __deepcopy__(self, memo):          # events / pinned buffers are per-instance runtime state, never copied
        return GradScaleTracker()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self.__init__()

    def observe(self, dout):
        if LP() != torch.float16 or _GRAD_SCALE_PINNED or torch.cuda.is_current_stream_capturing():
            return
        if self._host is None:
            self._host = torch.zeros(2, dtype=torch.float32).pin_memory()
        if self._seen - self._used >= 2:   # two backwards without a training forward in between (gradient accumulation):
            self._used = self._seen - 1    # the oldest unread observation is dropped, by count -- still timing-independent
        slot = self._seen % 2
        self._host[slot:slot + 1].copy_(torch.linalg.vector_norm(dout.reshape(-1), float("inf")).reshape(1), non_blocking=True)
        ev = self._events[slot] or torch.cuda.Event()
        ev.record()
        self._events[slot] = ev
        self._seen += 1

    def update(self):
        global _GRAD_SCALE
        if self._host is None or _GRAD_SCALE_PINNED:
            return
        if self._seen - self._used < 2:    # the observation of the step before last is the newest one that is used
            return
        slot = self._used % 2
        self._events[slot].synchronize()
        self._used += 1
        amax = float(self._host[slot])
        if amax > 0.0 and amax == amax and amax != float("inf"):
            import math
            k = min(24, max(0, math.floor(math.log2(self.TARGET / amax))))
            _GRAD_SCALE = float(2 ** k)


# optional per-launch timing (bench.py): an object with .add(name, ev_start, ev_end, flops, bytes)
_timer = None


def set_timer(t):
    global _timer
    _timer = t


class KernelTimer:
    """HIP-event brackets around individual kernel launches, recorded on the launch stream."""

    def __init__(self, only=None):
        self.rec, self.only = [], only

    def start(self, name):
        if self.only is not None and name not in self.only:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, name, e0, flops=0.0, nbytes=0.0):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.rec.append((name, e0, e1, flops, nbytes))

    def summary(self):
        out = {}
        for name, e0, e1, fl, nb in self.rec:
            d = out.setdefault(name, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += nb
        return out


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def _stream():
    """the calling thread's current HIP stream as a raw handle.  `torch.cuda.current_stream().cuda_stream` builds a Stream
    object through five Python frames (8.5 us; ~465 kernel launches per training step = 4 ms of host time per step); the C
    accessor behind it costs 0.3 us."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise _lib.SimvgHipError(f"{name} must live in HBM (got a CPU tensor): simvg_amd has no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")
    return t


def gemm_nt(a, w, bias=None, out=None, out_dtype=None, split=0, act=0, aux_preact=None, residual=None,
            row_scale=None, rows_per_sample=(1, 1), w_group_stride=None, bias_group_stride=None, alpha=1.0):
    """out[M,N] = a[M,K] @ w[g][N,K]^T (+bias) (+act) (+residual + row_scale*...).  w: [N,K] or [2,N,K]."""
    lib = _lib.load()
    _chk(a, LP(), "a"); _chk(w, LP(), "w")
    M, K = a.shape
    N = w.shape[-2]
    assert w.shape[-1] == K
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=out_dtype or LP())
    if w_group_stride is None:
        w_group_stride = w.stride(0) if w.dim() == 3 else 0
    if bias is not None:
        _chk(bias, torch.float32, "bias")
        if bias_group_stride is None:
            bias_group_stride = bias.stride(0) if bias.dim() == 2 else 0
    tname = "gemm_nt"
    if _timer is not None and _timer.only is None:      # --breakdown: one line per shape / epilogue
        tname = f"gemm_nt[{M}x{N}x{K}{'+res' if residual is not None else ''}{'+f32' if out.dtype == torch.float32 else ''}]"
    t0 = _timer.start(tname) if _timer is not None else None
    rc = lib.simvg_gemm_nt(_p(a), a.stride(0), _p(w), w_group_stride, w.stride(-2), _p(bias), bias_group_stride or 0,
                           _p(out), out.stride(0), int(out.dtype == torch.float32),
                           _p(aux_preact), aux_preact.stride(0) if aux_preact is not None else 0,
                           _p(residual), residual.stride(0) if residual is not None else 0,
                           _p(row_scale), rows_per_sample[0], rows_per_sample[1], M, N, K, split, act, alpha, _stream())
    if t0 is not None:
        _timer.stop(tname, t0, 2.0 * M * N * K, 2.0 * (M * K + N * K) + out.element_size() * M * N)
    _lib.check(rc, "simvg_gemm_nt")
    return out


SPLIT_SHIFT = 11        # lo = 16-bit rounding of (w - hi) * 2^11: the same magnitude as w's own rounding step, normal range


def split_weight(w32, out=None):
    """fp32 weight [..., N, K] -> 16-bit [..., N, 2 K] = [lo * 2^SPLIT_SHIFT | hi] (operand of `gemm_nt_split`)"""
    hi = w32.to(LP())
    lo = ((w32 - hi.float()) * float(2 ** SPLIT_SHIFT)).to(LP())
    if out is None:
        return torch.cat([lo, hi], dim=-1)
    K = w32.shape[-1]
    out[..., :K].copy_(lo)
    out[..., K:].copy_(hi)
    return out


def gemm_nt_split(a, w2, bias=None, out=None, out_dtype=None, split=0, residual=None):
    """out[M,N] = a[M,K] @ (hi + lo)[g][N,K]^T (+bias) (+residual); w2 = split_weight(w): [N, 2K] or [2, N, 2K]"""
    lib = _lib.load()
    _chk(a, LP(), "a"); _chk(w2, LP(), "w2")
    M, K = a.shape
    N = w2.shape[-2]
    assert w2.shape[-1] == 2 * K
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=out_dtype or LP())
    if bias is not None:
        _chk(bias, torch.float32, "bias")
    # the roofline family "gemm_nt" counts this launch with the ALGORITHMIC work of the Linear it computes (2 M N K; the kernel executes
    # twice that on the MFMA pipe): the price of the hi + lo weights shows in the family's fraction instead of hiding beside it
    tname = "gemm_nt"
    if _timer is not None and _timer.only is None:      # --breakdown: one line per shape
        tname = f"gemm_nt_split[{M}x{N}x{K}{'+res' if residual is not None else ''}{'+f32' if out.dtype == torch.float32 else ''}]"
    t0 = _timer.start(tname) if _timer is not None else None
    rc = lib.simvg_gemm_nt_split(_p(a), a.stride(0), _p(w2), w2.stride(0) if w2.dim() == 3 else 0, w2.stride(-2), _p(bias),
                                 (bias.stride(0) if bias.dim() == 2 else 0) if bias is not None else 0,
                                 _p(out), out.stride(0), int(out.dtype == torch.float32),
                                 _p(residual), residual.stride(0) if residual is not None else 0,
                                 M, N, K, split, float(2.0 ** -SPLIT_SHIFT), _stream())
    if t0 is not None:
        _timer.stop(tname, t0, 2.0 * M * N * K, 2.0 * (M * K + 2 * N * K) + out.element_size() * M * N)
    _lib.check(rc, "simvg_gemm_nt_split")
    return out


_tn_ws = {}


class WgradReduceBatch:
    """Second stages of several `gemm_tn(..., defer=batch)` calls (the sum of the row partitions' slabs into dW) as ONE launch
    (`simvg_wgrad_reduce_batched`): every deferred call gets a slab workspace of its own from the batch's pool (it has to
    survive until `flush()`), `flush()` launches the batched reduction on the current stream and frees the pool for reuse.
    (Measured in situ, profiles/r04_sweeps.md: one in-line launch per encoder layer 30.6 ms per step, the same launch on a side
    stream beside the next layer's kernels 31.0 -- it competes with the HBM-bound LayerNorm kernels --, fp32 atomics 31.1.)"""

    def __init__(self):
        import ctypes as C
        self._C = C
        self.descs = (_lib.WgradReduceDesc * 16)()
        self.n = 0
        self._pool = {}
        self._used = {}

    def workspace(self, nfloats, device):
        key = (str(device), nfloats)
        k = self._used.get(key, 0)
        pool = self._pool.setdefault(key, [])
        if k == len(pool):
            pool.append(torch.empty(nfloats, device=device, dtype=torch.float32))
        self._used[key] = k + 1
        return pool[k]

    def next_desc(self):
        if self.n == len(self.descs):
            self.flush()
        return self._C.byref(self.descs[self.n])

    def commit(self):
        self.n += 1

    def reset(self):
        """drop whatever an aborted backward left behind (descriptors of slabs that were never reduced)"""
        self.n = 0
        self._used = {}

    def flush(self):
        if self.n:
            t0 = _timer.start("wgrad_reduce") if _timer is not None else None
            rc = _lib.load().simvg_wgrad_reduce_batched(self._C.byref(self.descs), self.n, _stream())
            if t0 is not None:      # reads every partition's slab and dW, writes dW
                nb = sum(4.0 * d.N * d.K * ((d.hi0 - d.lo0) + (d.hi1 - d.lo1) + (1 if d.assign else 2) * ((d.hi0 > d.lo0) + (d.hi1 > d.lo1)))
                         for d in self.descs[:self.n])
                _timer.stop("wgrad_reduce", t0, 0.0, nb)
            _lib.check(rc, "simvg_wgrad_reduce_batched")
        self.n = 0
        self._used = {}


def gemm_tn_can_assign(M, N, K):
    """True if `gemm_tn(..., defer=batch, assign=True)` is available for this problem: it runs on a kernel that leaves its partial
    sums in slabs (csrc/wgrad.hip), so the batched second stage can WRITE dw instead of adding to it."""
    return int(_lib.load().simvg_gemm_tn_ws_floats(M, N, K)) > 0


def gemm_tn(dy, x, dw, split=0, dw_group_stride=None, db=None, out_scale=1.0, defer=None, assign=False):
    """dw[g][N,K] += dy[M,N]^T @ x[M,K]   (fp32 accumulate into dw); db[g][N] += column sums of dy (optional).
    defer: a `WgradReduceBatch` -- dw is complete only after the batch's `flush()`.
    assign (with defer, where `gemm_tn_can_assign`): dw = ... instead of dw += ...: dw need not be zeroed and is not read (db still
    accumulates)."""
    lib = _lib.load()
    _chk(dy, LP(), "dy"); _chk(x, LP(), "x"); _chk(dw, torch.float32, "dw")
    M, N = dy.shape
    K = x.shape[1]
    if dw_group_stride is None:
        dw_group_stride = dw.stride(0) if dw.dim() == 3 else 0
    tname = f"gemm_tn[{M}x{N}x{K}]" if _timer is not None and _timer.only is None else "gemm_tn"
    t0 = _timer.start(tname) if _timer is not None else None
    nws = int(lib.simvg_gemm_tn_ws_floats(M, N, K))
    if nws:          # slabs for the partial sums of the XCD-partitioned kernel
        if defer is not None:
            # the descriptor first: a full table flushes here and frees the pool, BEFORE this call's slabs are taken from it
            dptr = defer.next_desc()
            ws = defer.workspace(nws, dy.device)
        else:
            key = (str(dy.device), nws)
            ws = _tn_ws.get(key)
            if ws is None:
                ws = _tn_ws[key] = torch.empty(nws, device=dy.device, dtype=torch.float32)
            dptr = None
        rc = lib.simvg_gemm_tn_ws(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(dw), dw_group_stride, dw.stride(-2),
                                  _p(db), (db.stride(0) if db.dim() == 2 else 0) if db is not None else 0,
                                  M, N, K, split, out_scale, _p(ws), dptr, _stream())
        if defer is not None:
            _lib.check(rc, "simvg_gemm_tn_ws")          # a refused call wrote no descriptor: never commit a stale one
            if assign:
                if not defer.descs[defer.n].slabs:
                    raise RuntimeError("gemm_tn(assign=True): this problem has no second stage (check gemm_tn_can_assign)")
                defer.descs[defer.n].assign = 2 if (dw.dim() == 3 and dw.shape[0] == 2) else 1     # 2: a group without rows is zeroed
            defer.commit()
        elif assign:
            raise ValueError("gemm_tn(assign=True) needs defer=")
    else:
        if assign:
            raise RuntimeError("gemm_tn(assign=True): this problem has no second stage (check gemm_tn_can_assign)")
        rc = lib.simvg_gemm_tn(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(dw), dw_group_stride, dw.stride(-2),
                               _p(db), (db.stride(0) if db.dim() == 2 else 0) if db is not None else 0,
                               M, N, K, split, out_scale, _stream())
    if t0 is not None:
        _timer.stop(tname, t0, 2.0 * M * N * K, 2.0 * M * (N + K) + 4.0 * N * K)
    _lib.check(rc, "simvg_gemm_tn")
    return dw


def colsum(y, out, split=0, out_group_stride=None):
    lib = _lib.load()
    _chk(y, LP(), "y"); _chk(out, torch.float32, "out")
    M, N = y.shape
    if out_group_stride is None:
        out_group_stride = out.stride(0) if out.dim() == 2 else 0
    t0 = _timer.start("colsum") if _timer is not None else None
    rc = lib.simvg_colsum(_p(y), y.stride(0), _p(out), out_group_stride, M, N, split, _stream())
    if t0 is not None:
        _timer.stop("colsum", t0, 0.0, 2.0 * M * N)
    _lib.check(rc, "simvg_colsum")
    return out


def ln_fwd(x, gamma, beta, split=0, eps=1e-5, out_lp=True, out_f32=False, save_stats=True, y=None, y32=None,
           gelu_in=False):
    """gamma/beta: [D] or [2,D] fp32.  Returns (y_bf16|None, y_f32|None, mean, rstd).  gelu_in: x is the fc1
    pre-activation, LayerNorm(gelu(x)) is computed."""
    lib = _lib.load()
    _chk(x, None, "x")
    M, D = x.shape
    gs = gamma.stride(0) if gamma.dim() == 2 else 0
    if out_lp and y is None:
        y = torch.empty(M, D, device=x.device, dtype=LP())
    if out_f32 and y32 is None:
        y32 = torch.empty(M, D, device=x.device, dtype=torch.float32)
    mean = torch.empty(M, device=x.device, dtype=torch.float32) if save_stats else None
    rstd = torch.empty(M, device=x.device, dtype=torch.float32) if save_stats else None
    t0 = _timer.start("ln_fwd") if _timer is not None else None
    rc = lib.simvg_ln_fwd(_p(x), int(x.dtype == LP()), x.stride(0), _p(gamma), _p(beta), gs, _p(y),
                          y.stride(0) if y is not None else 0, _p(y32), y32.stride(0) if y32 is not None else 0,
                          _p(mean), _p(rstd), M, D, split, eps, int(gelu_in), _stream())
    if t0 is not None:
        _timer.stop("ln_fwd", t0, 0.0, float(M) * D * (x.element_size() + (2 if y is not None else 0) + (4 if y32 is not None else 0)))
    _lib.check(rc, "simvg_ln_fwd")
    return y, y32, mean, rstd


_ln_ws = {}


def _ln_workspace(M, D, split, device):
    n = int(_lib.load().simvg_ln_bwd_ws_floats(M, D, split))
    key = (str(device), n)
    t = _ln_ws.get(key)
    if t is None:
        t = torch.empty(n, device=device, dtype=torch.float32)
        _ln_ws[key] = t
    return t


class LnReduceBatch:
    """The second stages of several `ln_bwd(..., defer=batch)` calls as ONE launch (`simvg_ln_param_reduce_batched`): each
    deferred call gets a partial workspace of its own from a per-batch pool (it has to survive until `flush()`), its
    description is kept on the host, `flush()` launches them all.  Same sums in the same order as the per-call second stage."""

    def __init__(self):
        import ctypes as C
        self._C = C
        self.descs = (_lib.LnReduceDesc * 64)()
        self.n = 0
        self._pool = {}          # (device, floats) -> [workspaces]; `_used` counts how many of each are taken this batch
        self._used = {}

    def workspace(self, M, D, split, device):
        n = int(_lib.load().simvg_ln_bwd_ws_floats(M, D, split))
        key = (str(device), n)
        k = self._used.get(key, 0)
        pool = self._pool.setdefault(key, [])
        if k == len(pool):
            pool.append(torch.empty(n, device=device, dtype=torch.float32))
        self._used[key] = k + 1
        return pool[k]

    def next_desc(self):
        if self.n == len(self.descs):
            self.flush()
        return self._C.byref(self.descs[self.n])

    def commit(self):
        self.n += 1

    def reset(self):
        """drop whatever an aborted backward left behind"""
        self.n = 0
        self._used = {}

    def flush(self):
        if self.n:
            rc = _lib.load().simvg_ln_param_reduce_batched(self._C.byref(self.descs), self.n, _stream())
            _lib.check(rc, "simvg_ln_param_reduce_batched")
        self.n = 0
        self._used = {}


def ln_bwd(dy, x, mean, rstd, gamma, dgamma, dbeta, split=0, dx_lp=None, gelu_u=None, dres=None, dx_f32=None,
           dx_scaled=None, row_scale=None, rows_per_sample=(1, 1), dy_scale=1.0, param_scale=1.0, defer=None):
    """defer: a `LnReduceBatch` -- the second stage of the two-stage dgamma / dbeta reduction joins the batch's single launch
    (dgamma / dbeta are complete only after `defer.flush()`)"""
    lib = _lib.load()
    # two-stage dgamma/dbeta reduction pays for wide rows only (measured: profiles/r01_sweeps.md)
    # (and for the many-rows-in-flight kernel of the 768 / 1024-wide 16-bit-dy instances, csrc/layernorm.hip: ln_bwd_tile_kernel)
    ws = dptr = None
    two_stage = dy.shape[0] >= 1024 and (dy.shape[1] >= 2048 or (dy.shape[1] in (768, 1024) and dy.dtype != torch.float32))
    if two_stage:
        if defer is not None:
            # the descriptor first: a full table flushes here and frees the pool, BEFORE this call's workspace is taken from it
            dptr = defer.next_desc()
            ws = defer.workspace(dy.shape[0], dy.shape[1], split, dy.device)
        else:
            ws = _ln_workspace(dy.shape[0], dy.shape[1], split, dy.device)
    _chk(dy, None, "dy")
    M, D = dy.shape
    gs = gamma.stride(0) if gamma.dim() == 2 else 0
    t0 = _timer.start("ln_bwd") if _timer is not None else None
    args = (_p(dy), int(dy.dtype == torch.float32), dy.stride(0), _p(x), int(x.dtype == LP()), x.stride(0), _p(mean), _p(rstd),
            _p(gamma), gs, _p(dgamma), _p(dbeta), _p(dx_lp),
            dx_lp.stride(0) if dx_lp is not None else 0, _p(gelu_u),
            gelu_u.stride(0) if gelu_u is not None else 0, _p(dres), _p(dx_f32),
            dx_f32.stride(0) if dx_f32 is not None else 0, _p(dx_scaled),
            dx_scaled.stride(0) if dx_scaled is not None else 0, _p(row_scale),
            rows_per_sample[0], rows_per_sample[1], M, D, split, _p(ws), dy_scale, param_scale)
    if defer is not None and two_stage:
        rc = lib.simvg_ln_bwd_deferred(*args, dptr, _stream())
        _lib.check(rc, "simvg_ln_bwd_deferred")
        defer.commit()
    else:
        rc = lib.simvg_ln_bwd(*args, _stream())
    if t0 is not None:
        nb = dy.element_size() + x.element_size() + (2 if dx_lp is not None else 0) + (2 if gelu_u is not None else 0) \
            + (4 if dres is not None else 0) + (4 if dx_f32 is not None else 0) + (2 if dx_scaled is not None else 0)
        _timer.stop("ln_bwd", t0, 0.0, float(M) * D * nb)
    _lib.check(rc, "simvg_ln_bwd")


def _philox_key():
    """A fresh 64-bit Philox key per call from torch's default CPU generator (a host-side draw, no device launch): `torch.manual_seed`
    / `set_random_seed` make the masks reproducible and ranks seeded differently draw different masks, exactly as with the
    framework's own dropout."""
    hi, lo = torch.randint(0, 1 << 31, (2,)).tolist()
    return (hi << 31) | lo


_philox_epochs = {}


def dropout_mult(n, device, keep=None, keep_seg=None, seg=0, out=None):
    """n multipliers, 0 or 1 / keep (csrc/rng.hip, Philox4x32-10).  keep: scalar keep probability; keep_seg: device tensor of
    keep probabilities, element i uses keep_seg[i // seg].  While the current stream is being captured into a hipGraph the
    launch gets a device-side epoch (one per device, owned by the training stream), so replays draw fresh multipliers."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(n, device=device, dtype=torch.float32)
    state = None
    if torch.cuda.is_current_stream_capturing():
        state = _philox_epochs.get(out.device)
        if state is None:
            raise RuntimeError("dropout_mult: first call on this device happens inside a graph capture; run one eager step first")
    elif out.device not in _philox_epochs:
        _philox_epochs[out.device] = torch.zeros(2, device=out.device, dtype=torch.int64)
    rc = lib.simvg_dropout_mult(_p(out), n, float(keep if keep is not None else 1.0), _p(keep_seg), int(seg), _philox_key(), 0,
                                _p(state), _stream())
    _lib.check(rc, "simvg_dropout_mult")
    return out


def attn_fwd(qkv, B, H, Nv, Nt, pad=None, out=None, scale=None):
    """qkv: [M, 3*D] lp, modality-major rows.  Returns (out [M,D] lp, lse [B*H, N] fp32)."""
    lib = _lib.load()
    _chk(qkv, LP(), "qkv")
    M, D3 = qkv.shape
    D = D3 // 3
    N = Nv + Nt
    if out is None:
        out = torch.empty(M, D, device=qkv.device, dtype=LP())
    lse = torch.empty(B * H, N, device=qkv.device, dtype=torch.float32)
    if scale is None:
        scale = (D // H) ** -0.5
    t0 = _timer.start("attn_fwd") if _timer is not None else None
    rc = lib.simvg_attn_fwd(_p(qkv), qkv.stride(0), _p(out), out.stride(0), _p(lse), _p(pad), B, H, Nv, Nt, D,
                            scale, _stream())
    if t0 is not None:
        _timer.stop("attn_fwd", t0, 4.0 * B * H * N * N * (D // H), 2.0 * M * 4 * D)
    _lib.check(rc, "simvg_attn_fwd")
    return out, lse


def attn_qk_probe(qkv, B, H, Nv, Nt, pad=None, scale=None):
    """the QK^T contraction of `attn_fwd` alone (measurement): row maxima of the scaled scores [B*H, N]"""
    lib = _lib.load()
    _chk(qkv, LP(), "qkv")
    D = qkv.shape[1] // 3
    rowmax = torch.empty(B * H, Nv + Nt, device=qkv.device, dtype=torch.float32)
    rc = lib.simvg_attn_qk_probe(_p(qkv), qkv.stride(0), _p(rowmax), _p(pad), B, H, Nv, Nt, D, (D // H) ** -0.5 if scale is None else scale, _stream())
    _lib.check(rc, "simvg_attn_qk_probe")
    return rowmax


def attn_bwd(qkv, out, dout, lse, B, H, Nv, Nt, pad=None, dqkv=None, scale=None):
    lib = _lib.load()
    M, D3 = qkv.shape
    D = D3 // 3
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    if scale is None:
        scale = (D // H) ** -0.5
    t0 = _timer.start("attn_bwd") if _timer is not None else None
    rc = lib.simvg_attn_bwd(_p(qkv), qkv.stride(0), _p(out), out.stride(0), _p(dout), dout.stride(0), _p(dqkv),
                            dqkv.stride(0), _p(lse), _p(delta), _p(pad), B, H, Nv, Nt, D, scale, _stream())
    if t0 is not None:
        _timer.stop("attn_bwd", t0, 10.0 * B * H * (Nv + Nt) ** 2 * (D // H), 2.0 * M * 8 * D)
    _lib.check(rc, "simvg_attn_bwd")
    return dqkv


def im2col(img, P, out=None):
    lib = _lib.load()
    _chk(img, torch.float32, "img")
    assert img.is_contiguous()
    B, Cc, S, S2 = img.shape
    assert Cc == 3 and S == S2
    if out is None:
        out = torch.empty(B * (S // P) ** 2, 3 * P * P, device=img.device, dtype=LP())
    _lib.check(lib.simvg_im2col(_p(img), _p(out), B, S, P, _stream()), "simvg_im2col")
    return out


def embed_fwd(patch, cls, posA, posB, text_embed, ids, pad, B, np_, T, x=None):
    lib = _lib.load()
    D = patch.shape[1]
    if x is None:
        x = torch.empty(B * (np_ + 1 + T), D, device=patch.device, dtype=torch.float32)
    rc = lib.simvg_embed_fwd(_p(patch), patch.stride(0), _p(cls), _p(posA), _p(posB), _p(text_embed), _p(ids),
                             _p(pad), _p(x), x.stride(0), B, np_, T, D, _stream())
    _lib.check(rc, "simvg_embed_fwd")
    return x


def embed_bwd(dx, dpatch, dcls, dposA, dposB, dtext, ids, pad, B, np_, T, param_scale=1.0):
    lib = _lib.load()
    D = dx.shape[1]
    rc = lib.simvg_embed_bwd(_p(dx), dx.stride(0), _p(dpatch), dpatch.stride(0), _p(dcls), _p(dposA), _p(dposB),
                             _p(dtext), _p(ids), _p(pad), B, np_, T, D, param_scale, _stream())
    _lib.check(rc, "simvg_embed_bwd")


def cast_lp(src, dst=None, scale=1.0):
    """fp32 -> the 16-bit format (times `scale`, saturating)"""
    lib = _lib.load()
    if dst is None:
        dst = torch.empty(src.shape, device=src.device, dtype=LP())
    _lib.check(lib.simvg_cast_f32_to_lp(_p(src), _p(dst), src.numel(), scale, _stream()), "simvg_cast_f32_to_lp")
    return dst


class WeightPrep:
    """Batched fp32 -> 16-bit (+ transposed) conversion of many weight matrices in ONE launch."""

    _generations = 0

    def __init__(self, entries, device):
        # entries: list of (src fp32 2-D tensor, dst lp | None, dst_t lp | None[, split_shift]); split_shift > 0: dst is [rows, 2 cols] =
        # [lo * 2^shift | hi] (the operand of `gemm_nt_split`; its right half is the plain 16-bit copy)
        # generation: a process-wide serial number -- what captured graphs key on to know which set of 16-bit buffers their
        # launches point into (an id() can be handed to a later object)
        WeightPrep._generations += 1
        self.generation = WeightPrep._generations
        n = len(entries)
        arr = (_lib.WeightDesc * n)()
        tiles = 0
        self._keep = entries
        for i, ent in enumerate(entries):
            src, dst, dst_t = ent[:3]
            shift = ent[3] if len(ent) > 3 else 0
            rows, cols = src.shape
            assert src.is_contiguous() and (dst is None or dst.is_contiguous()) and (dst_t is None or dst_t.is_contiguous())
            assert shift == 0 or (dst is not None and tuple(dst.shape) == (rows, 2 * cols))
            arr[i] = _lib.WeightDesc(src.data_ptr(), dst.data_ptr() if dst is not None else None,
                                     dst_t.data_ptr() if dst_t is not None else None, rows, cols, tiles, shift)
            tiles += ((rows + 63) // 64) * ((cols + 63) // 64)
        raw = bytes(arr)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.n, self.tiles = n, tiles

    def run(self):
        lib = _lib.load()
        _lib.check(lib.simvg_weight_prep(_p(self.table), self.n, self.tiles, _stream()), "simvg_weight_prep")


# ---------------------------------------------------------------------------------------------
# decoder head: exact-fp32 small GEMM, small attention, matcher, criterion
# ---------------------------------------------------------------------------------------------
def gemm_f32(A, sam, sak, Bm, sbk, sbn, C, M, N, K, bias=None, addend=None, addend_rows=0, accumulate=False, act=0):
    """C[M,N] (+)= sum_k A(m,k) B(k,n) with explicit element strides (pointers may be offset views)."""
    lib = _lib.load()
    name = f"gemm_f32[{M}x{N}x{K}]" if _timer is not None and _timer.only is None else "gemm_f32"
    e0 = _timer.start(name) if _timer is not None else None
    rc = lib.simvg_gemm_f32(_p(A), sam, sak, _p(Bm), sbk, sbn, _p(C), C.stride(0), _p(bias), _p(addend),
                            addend.stride(0) if addend is not None else 0, addend_rows, M, N, K, int(accumulate), act,
                            _stream())
    _lib.check(rc, "simvg_gemm_f32")
    if e0 is not None:
        _timer.stop(name, e0, 2.0 * M * N * K, 4.0 * (M * K + N * K + M * N))
    return C


def gemm_f32_group(problems):
    """Independent gemm_f32 problems in ONE launch.  Each problem: dict(A, sam, sak, B, sbk, sbn, C, M, N, K[, bias,
    addend, addend_rows, accumulate, act, A2, B2, mult, gate]) with the meaning of `gemm_f32`; A2 / B2: second operands
    (same strides) added to A / B on load; mult: [M, N] factor after the activation; gate: [M, N], the value passes where
    gate > 0 (with mult or gate the addend is added after them).  At most 12 per launch (longer lists are split)."""
    lib = _lib.load()
    for i in range(0, len(problems), 12):
        chunk = problems[i:i + 12]
        arr = (_lib.GemmF32Problem * len(chunk))()
        for d, q in zip(arr, chunk):
            ad = q.get("addend")
            d.A, d.sam, d.sak = q["A"].data_ptr(), q["sam"], q["sak"]
            d.B, d.sbk, d.sbn = q["B"].data_ptr(), q["sbk"], q["sbn"]
            d.C, d.ldc = q["C"].data_ptr(), q["C"].stride(0)
            b = q.get("bias")
            d.bias = b.data_ptr() if b is not None else None
            d.addend = ad.data_ptr() if ad is not None else None
            d.ld_addend = ad.stride(0) if ad is not None else 0
            d.addend_rows = q.get("addend_rows", 0)
            d.M, d.N, d.K = q["M"], q["N"], q["K"]
            d.accumulate, d.act = int(q.get("accumulate", False)), q.get("act", 0)
            a2, b2, mu, ga = q.get("A2"), q.get("B2"), q.get("mult"), q.get("gate")
            d.A2 = a2.data_ptr() if a2 is not None else None
            d.B2 = b2.data_ptr() if b2 is not None else None
            d.mult, d.ld_mult = (mu.data_ptr(), mu.stride(0)) if mu is not None else (None, 0)
            d.gate, d.ld_gate = (ga.data_ptr(), ga.stride(0)) if ga is not None else (None, 0)
        t0 = _timer.start("gemm_f32_group") if _timer is not None else None
        stream = _stream()
        ws = _group_workspace(chunk[0]["C"].device, stream) if any(q["M"] * q["N"] * q["K"] >= 40e6 for q in chunk) else None
        rc = lib.simvg_gemm_f32_grouped_ws(C.byref(arr), len(chunk), _p(ws), 0 if ws is None else ws.numel(), stream)
        if t0 is not None:
            _timer.stop("gemm_f32_group", t0, sum(2.0 * q["M"] * q["N"] * q["K"] for q in chunk), 0.0)
        _lib.check(rc, "simvg_gemm_f32_grouped")


_GROUP_WS = {}
_GROUP_WS_FLOATS = 16 << 20        # 64 MB: twelve problems' split-K slabs at num_queries = 10 (the largest: 13 MB)


def _group_workspace(device, stream):
    """split-K slabs of `gemm_f32_group` (one buffer per device and stream: launches of one stream are ordered, two streams
    must not share it)"""
    key = (device.index, stream.value)
    ws = _GROUP_WS.get(key)
    if ws is None:
        ws = _GROUP_WS[key] = torch.empty(_GROUP_WS_FLOATS, device=device, dtype=torch.float32)
    return ws


def gp(A, sam, sak, Bm, sbk, sbn, Cm, M, N, K, **kw):
    """one problem of `gemm_f32_group` (same positional arguments as `gemm_f32`)"""
    return dict(A=A, sam=sam, sak=sak, B=Bm, sbk=sbk, sbn=sbn, C=Cm, M=M, N=N, K=K, **kw)


def attn_small_fwd(q, k, v, B, H, Lq, Lk, kpm=None, drop=None, kv_rows=0, kpos=None):
    """kpos: projected key_pos rows [Lk, E] (shared) or [B * Lk, E], added to the K rows on load"""
    lib = _lib.load()
    E = H * 32
    out = torch.empty(B * Lq, E, device=q.device, dtype=torch.float32)
    P = torch.empty(B, H, Lq, Lk, device=q.device, dtype=torch.float32)
    rc = lib.simvg_attn_small_fwd(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(out), out.stride(0),
                                  _p(P), _p(kpm), _p(drop), B, H, Lq, Lk, kv_rows, 32 ** -0.5, *_kpos_args(kpos, Lk), _stream())
    _lib.check(rc, "simvg_attn_small_fwd")
    return out, P


def _kpos_args(kpos, Lk):
    if kpos is None:
        return None, 0, 0
    assert kpos.dim() == 2 and kpos.stride(1) == 1 and kpos.shape[0] % Lk == 0
    return _p(kpos), kpos.stride(0), (Lk if kpos.shape[0] > Lk else 0)


def attn_small_bwd(q, k, v, P, dout, dq, dk, dv, B, H, Lq, Lk, kpm=None, drop=None, kv_rows=0, kpos=None):
    lib = _lib.load()
    rc = lib.simvg_attn_small_bwd(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(P), _p(kpm), _p(drop),
                                  _p(dout), dout.stride(0), _p(dq), dq.stride(0), _p(dk), dk.stride(0), _p(dv),
                                  dv.stride(0), B, H, Lq, Lk, kv_rows, 32 ** -0.5, *_kpos_args(kpos, Lk), _stream())
    _lib.check(rc, "simvg_attn_small_bwd")


# ---------------------------------------------------------------------------------------------------------------------
# decoder layers as few launches (csrc/decoder.hip)
DEC_E, DEC_H = 256, 8
_DEC_SAVED = ("qkv", "P0", "o", "r1", "mean1", "rstd1", "t1", "qc", "qk", "P1", "ctx", "sp", "o2", "r2", "mean2", "rstd2", "t2")


def _carve(sizes, device):
    """one allocation, 16-byte aligned fp32 views of the given element counts"""
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + 3) // 4 * 4
    flat = torch.empty(total, device=device, dtype=torch.float32)
    return [flat[o:o + n] for o, n in zip(offs, sizes)]


def dec_attn_fwd(tgt, qpos, W, src, B, R, Lk, kv_rows=0, kv_off=0, kpos=None, kpm=None, dm0=None, dm1=None, eps=1e-5):
    """The attention block of one decoder layer (self-attention, norm, cross-attention, norm) in one launch.
    tgt, qpos [B*R, E] fp32; W = (Ws, bs, Wso, bso, g0, b0, Wc, bc, Wco, bco, g1, b1) (the layer's parameters, reference layout);
    src: the cross-attention's source rows [B*kv_rows, E], 16-bit (LP()) or fp32 -- keys are src + kpos, values are src; sample
    b uses rows b*kv_rows + kv_off + [0, Lk).  kpos [Lk, E] (shared) or [B*Lk, E]; kpm [B, Lk] uint8; dm0 / dm1 dropout
    multipliers [B,H,R,R] / [B,H,R,Lk].  Returns the dict of saved tensors; ["t2"] is the block's output [B*R, E]."""
    lib = _lib.load()
    E, H, M = DEC_E, DEC_H, B * R
    dev = qpos.device
    for t, n in ((tgt, "tgt"), (qpos, "qpos")):
        if t is None and n == "tgt":          # None: zeros (the first layer of a decoder) -- no buffer is filled or read
            continue
        _chk(t, torch.float32, n)
        assert tuple(t.shape) == (M, E) and t.is_contiguous(), n
    for w in W:
        _chk(w, torch.float32, "decoder parameter")
        assert w.is_contiguous()
    kv_rows = kv_rows or Lk
    assert src.dim() == 2 and src.shape[1] == E and src.stride(1) == 1 and src.shape[0] >= B * kv_rows
    sizes = dict(qkv=M * 3 * E, P0=B * H * R * R, o=M * E, r1=M * E, mean1=M, rstd1=M, t1=M * E, qc=M * E, qk=M * H * E,
                 P1=B * H * R * Lk, ctx=M * H * E, sp=M * H, o2=M * E, r2=M * E, mean2=M, rstd2=M, t2=M * E)
    bufs = dict(zip(_DEC_SAVED, _carve([sizes[k] for k in _DEC_SAVED], dev)))
    a = _lib.DecAttnArgs()
    a.B, a.R, a.Lk, a.kv_rows, a.kv_off = B, R, Lk, kv_rows, kv_off
    a.tgt, a.qpos = (tgt.data_ptr() if tgt is not None else None), qpos.data_ptr()
    for name, w in zip(("Ws", "bs", "Wso", "bso", "g0", "b0", "Wc", "bc", "Wco", "bco", "g1", "b1"), W):
        setattr(a, name, w.data_ptr())
    if src.dtype == LP():
        a.src16, a.src32 = src.data_ptr(), None
    else:
        _chk(src, torch.float32, "src")
        a.src16, a.src32 = None, src.data_ptr()
    a.ldsrc = src.stride(0)
    if kpos is not None:
        _chk(kpos, torch.float32, "kpos")
        assert kpos.dim() == 2 and kpos.stride(1) == 1 and kpos.shape[0] in (Lk, B * Lk)
        a.kpos, a.ldkp, a.kpos_rows = kpos.data_ptr(), kpos.stride(0), (Lk if kpos.shape[0] > Lk else 0)
    a.kpm = kpm.data_ptr() if kpm is not None else None
    a.dm0 = dm0.data_ptr() if dm0 is not None else None
    a.dm1 = dm1.data_ptr() if dm1 is not None else None
    for k in _DEC_SAVED:
        setattr(a, k, bufs[k].data_ptr())
    a.eps = eps
    t0 = _timer.start("dec_attn_fwd") if _timer is not None else None
    rc = lib.simvg_dec_attn_fwd(C.byref(a), _stream())
    if t0 is not None:
        _timer.stop("dec_attn_fwd", t0, 0.0, 0.0)
    _lib.check(rc, "simvg_dec_attn_fwd")
    shapes = dict(qkv=(M, 3 * E), P0=(B, H, R, R), P1=(B, H, R, Lk), qk=(M, H, E), ctx=(M, H, E), sp=(M, H), mean1=(M,), rstd1=(M,),
                  mean2=(M,), rstd2=(M,))
    return {k: v.view(shapes.get(k, (M, E))) for k, v in bufs.items()}


_DEC_BWD_ROWS = ("dt2sum", "gx2", "d_r2", "d_o2", "dqpre", "d_t1", "gx1", "d_r1")     # [M, E] each
_DEC_PARAM_GRADS = ("dWs", "dbs", "dWso", "dbso", "dg0", "db0", "dWc", "dbc", "dWco", "dbco", "dg1", "db1")


def dec_attn_bwd(saved, tgt, qpos, W, src, B, R, Lk, dt2=None, dt2_slabs=None, kv_rows=0, kv_off=0, kpos=None, dm0=None, dm1=None,
                 dsrc=None, dsrc_accumulate=False):
    """Backward of `dec_attn_fwd` (two launches).  saved: the forward's dict; dt2 [B*R, E] and / or dt2_slabs [n, B*R, E] (summed):
    the gradient of the block's output t2.  dsrc: fp32 [B*kv_rows, E] buffer for the gradient of the source rows (None: not
    computed); written, or added to with dsrc_accumulate.  Returns (d_tgt, d_qpos, {parameter gradients in W's order})."""
    lib = _lib.load()
    E, H, M = DEC_E, DEC_H, B * R
    dev = qpos.device
    kv_rows = kv_rows or Lk
    Ws, bs, Wso, bso, g0, b0, Wc, bc, Wco, bco, g1, b1 = W
    names = ("d_tgt", "d_qpos") + _DEC_BWD_ROWS + ("dctx", "dqk", "dqkv") + _DEC_PARAM_GRADS
    psize = dict(dWs=3 * E * E, dbs=3 * E, dWso=E * E, dbso=E, dg0=E, db0=E, dWc=3 * E * E, dbc=3 * E, dWco=E * E, dbco=E, dg1=E, db1=E)
    sizes = {n: M * E for n in ("d_tgt", "d_qpos") + _DEC_BWD_ROWS}
    sizes.update(dctx=M * H * E, dqk=M * H * E, dqkv=M * 3 * E, **psize)
    buf = dict(zip(names, _carve([sizes[n] for n in names], dev)))
    a = _lib.DecAttnBwdArgs()
    a.B, a.R, a.Lk, a.kv_rows, a.kv_off = B, R, Lk, kv_rows, kv_off
    for n, w in (("Ws", Ws), ("Wso", Wso), ("g0", g0), ("Wc", Wc), ("bc", bc), ("Wco", Wco), ("g1", g1)):
        setattr(a, n, w.data_ptr())
    if src.dtype == LP():
        a.src16, a.src32 = src.data_ptr(), None
    else:
        a.src16, a.src32 = None, src.data_ptr()
    a.ldsrc = src.stride(0)
    if kpos is not None:
        a.kpos, a.ldkp, a.kpos_rows = kpos.data_ptr(), kpos.stride(0), (Lk if kpos.shape[0] > Lk else 0)
    a.dm0 = dm0.data_ptr() if dm0 is not None else None
    a.dm1 = dm1.data_ptr() if dm1 is not None else None
    for n in ("qkv", "P0", "r1", "mean1", "rstd1", "qk", "P1", "r2", "mean2", "rstd2"):
        setattr(a, n, saved[n].data_ptr())
    if dt2 is not None:
        _chk(dt2, torch.float32, "dt2")
        assert dt2.is_contiguous() and dt2.numel() == M * E
        a.dt2 = dt2.data_ptr()
    if dt2_slabs is not None:
        _chk(dt2_slabs, torch.float32, "dt2_slabs")
        assert dt2_slabs.dim() == 3 and dt2_slabs.shape[1:] == (M, E) and dt2_slabs[0].is_contiguous()
        a.dt2_slabs, a.nslab, a.slab_stride = dt2_slabs.data_ptr(), dt2_slabs.shape[0], dt2_slabs.stride(0)
    a.d_tgt, a.d_qpos = buf["d_tgt"].data_ptr(), buf["d_qpos"].data_ptr()
    if dsrc is not None:
        _chk(dsrc, torch.float32, "dsrc")
        assert dsrc.dim() == 2 and dsrc.stride(1) == 1 and dsrc.shape[0] >= B * kv_rows and dsrc.shape[1] == E
        a.dsrc, a.lddsrc, a.dsrc_accumulate = dsrc.data_ptr(), dsrc.stride(0), int(bool(dsrc_accumulate))
    for n in _DEC_BWD_ROWS + ("dctx", "dqk", "dqkv"):
        setattr(a, n, buf[n].data_ptr())
    t0 = _timer.start("dec_attn_bwd") if _timer is not None else None
    rc = lib.simvg_dec_attn_bwd(C.byref(a), _stream())
    _lib.check(rc, "simvg_dec_attn_bwd")
    w = _lib.DecAttnWgradArgs()
    w.MR = M
    w.tgt, w.qpos = (tgt.data_ptr() if tgt is not None else None), qpos.data_ptr()
    for n in ("t1", "o", "o2", "ctx", "sp", "qc"):
        setattr(w, n, saved[n].data_ptr())
    for n in ("dqkv", "d_r1", "gx1", "d_t1", "dqpre", "dqk", "d_o2", "d_r2", "gx2", "dt2sum") + _DEC_PARAM_GRADS:
        setattr(w, n, buf[n].data_ptr())
    rc = lib.simvg_dec_attn_wgrad(C.byref(w), _stream())
    if t0 is not None:
        _timer.stop("dec_attn_bwd", t0, 0.0, 0.0)
    _lib.check(rc, "simvg_dec_attn_wgrad")
    pshape = dict(dWs=(3 * E, E), dWso=(E, E), dWc=(3 * E, E), dWco=(E, E))
    grads = [buf[n].view(pshape[n]) if n in pshape else buf[n] for n in _DEC_PARAM_GRADS]
    return buf["d_tgt"].view(M, E), buf["d_qpos"].view(M, E), grads


def dec_attn_max_keys():
    return int(_lib.load().simvg_dec_attn_max_keys())


def dec_ffn_fwd(t2, W1, b1, W2, b2, g2, b2n, gP=None, bP=None, m1=None, m2=None, eps=1e-5):
    """The FFN + norm (+ the decoder's post-norm) of one decoder layer in two launches; returns the dict of saved tensors
    (["t3"] the layer output, ["hs"] its post-norm or None)."""
    lib = _lib.load()
    M, E = t2.shape
    Fd = W1.shape[0]
    NS = Fd // 64
    dev = t2.device
    names = ("h1d", "slabs", "r3", "mean3", "rstd3", "t3") + (("hs", "meanP", "rstdP") if gP is not None else ())
    sizes = dict(h1d=M * Fd, slabs=NS * M * E, r3=M * E, mean3=M, rstd3=M, t3=M * E, hs=M * E, meanP=M, rstdP=M)
    buf = dict(zip(names, _carve([sizes[n] for n in names], dev)))
    a = _lib.DecFfnArgs()
    a.M, a.Fd = M, Fd
    a.t2, a.W1, a.b1, a.W2 = t2.data_ptr(), W1.data_ptr(), b1.data_ptr(), W2.data_ptr()
    a.m1 = m1.data_ptr() if m1 is not None else None
    a.h1d, a.slabs = buf["h1d"].data_ptr(), buf["slabs"].data_ptr()
    t0 = _timer.start("dec_ffn_fwd") if _timer is not None else None
    _lib.check(lib.simvg_dec_ffn_fwd(C.byref(a), _stream()), "simvg_dec_ffn_fwd")
    f = _lib.DecFfnFinishArgs()
    f.M, f.NS = M, NS
    f.t2, f.slabs, f.b2 = t2.data_ptr(), buf["slabs"].data_ptr(), b2.data_ptr()
    f.m2 = m2.data_ptr() if m2 is not None else None
    f.g2, f.b2n = g2.data_ptr(), b2n.data_ptr()
    f.gP, f.bP = (gP.data_ptr(), bP.data_ptr()) if gP is not None else (None, None)
    for n in ("r3", "mean3", "rstd3", "t3"):
        setattr(f, n, buf[n].data_ptr())
    if gP is not None:
        f.hs, f.meanP, f.rstdP = buf["hs"].data_ptr(), buf["meanP"].data_ptr(), buf["rstdP"].data_ptr()
    f.eps = eps
    _lib.check(lib.simvg_dec_ffn_finish(C.byref(f), _stream()), "simvg_dec_ffn_finish")
    if t0 is not None:
        _timer.stop("dec_ffn_fwd", t0, 0.0, 0.0)
    out = {n: (buf[n].view(M, Fd) if n == "h1d" else buf[n].view(NS, M, E) if n == "slabs" else
               buf[n] if n in ("mean3", "rstd3", "meanP", "rstdP") else buf[n].view(M, E)) for n in names}
    out.setdefault("hs", None)
    return out


def dec_ffn_bwd(saved, t2, W1, W2, g2, gP=None, d_t3=None, d_hs=None, m1=None, m2=None):
    """Backward of `dec_ffn_fwd`: returns (d_r3 [M, E], slabs [Fd/64, M, E] -- d(t2) = d_r3 + slabs.sum(0), formed by `dec_attn_bwd` --,
    [dW1, db1, dW2, db2, dg2, db2n, dgP | None, dbP | None])."""
    lib = _lib.load()
    M, E = t2.shape
    Fd = W1.shape[0]
    NS = Fd // 64
    dev = t2.device
    post = d_hs is not None
    names = ("d_r3", "gx3", "dy3", "dr3m", "slabs", "dW1", "db1", "dW2", "db2", "dg2", "db2n") + (("gxP", "dgP", "dbP") if post else ())
    sizes = dict(d_r3=M * E, gx3=M * E, dy3=M * E, dr3m=M * E, gxP=M * E, slabs=NS * M * E, dW1=Fd * E, db1=Fd, dW2=E * Fd, db2=E, dg2=E,
                 db2n=E, dgP=E, dbP=E)
    buf = dict(zip(names, _carve([sizes[n] for n in names], dev)))
    a = _lib.DecFfnBwdArgs()
    a.M, a.Fd = M, Fd
    for n, t in (("d_t3", d_t3), ("d_hs", d_hs)):
        if t is not None:
            _chk(t, torch.float32, n)
            assert t.is_contiguous() and t.numel() == M * E
            setattr(a, n, t.data_ptr())
    for n in ("r3", "mean3", "rstd3", "t3", "h1d"):
        setattr(a, n, saved[n].data_ptr())
    if post:
        a.meanP, a.rstdP, a.gP = saved["meanP"].data_ptr(), saved["rstdP"].data_ptr(), gP.data_ptr()
    a.g2, a.W1, a.W2, a.t2 = g2.data_ptr(), W1.data_ptr(), W2.data_ptr(), t2.data_ptr()
    a.m1 = m1.data_ptr() if m1 is not None else None
    a.m2 = m2.data_ptr() if m2 is not None else None
    for n in names:
        setattr(a, n, buf[n].data_ptr())
    t0 = _timer.start("dec_ffn_bwd") if _timer is not None else None
    rc = lib.simvg_dec_ffn_bwd(C.byref(a), _stream())
    if t0 is not None:
        _timer.stop("dec_ffn_bwd", t0, 0.0, 0.0)
    _lib.check(rc, "simvg_dec_ffn_bwd")
    grads = [buf["dW1"].view(Fd, E), buf["db1"], buf["dW2"].view(E, Fd), buf["db2"], buf["dg2"], buf["db2n"],
             buf["dgP"] if post else None, buf["dbP"] if post else None]
    return buf["d_r3"].view(M, E), buf["slabs"].view(NS, M, E), grads


class _PackRing:
    """pinned staging buffers for the one host->device copy of `pack_targets`: a slot is rewritten only after the copy that read it
    has completed (the host may run steps ahead of the device)"""
    SLOTS = 8

    def __init__(self):
        self.bufs, self.events, self.i = [None] * self.SLOTS, [None] * self.SLOTS, 0

    def stage(self, raw, device):
        k = self.i % self.SLOTS
        self.i += 1
        if self.events[k] is not None:
            self.events[k].synchronize()
        if self.bufs[k] is None or self.bufs[k].numel() < len(raw):
            self.bufs[k] = torch.empty(max(4096, 2 * len(raw)), dtype=torch.uint8).pin_memory()
        pin = self.bufs[k][:len(raw)]
        pin.copy_(torch.frombuffer(raw, dtype=torch.uint8))
        dev = pin.to(device, non_blocking=True)
        ev = self.events[k] or torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        return dev


_pack_ring = _PackRing()


def pack_targets(rows, counts, B, TM, device):
    """rows: list of (device tensor holding an fp32 xyxy box at `offset` elements | None, offset, (x0, y0, x1, y1) | None, w, h,
    destination row); counts: B ints.  -> (boxes [B, TM, 4] fp32 normalised cxcywh, count [B] int32), one copy + one launch."""
    import struct
    lib = _lib.load()
    raw = bytearray()
    for src, off, box, w, h, dst in rows:
        ptr = 0 if src is None else src.data_ptr() + 4 * off
        bx = box if box is not None else (0.0, 0.0, 0.0, 0.0)
        raw += struct.pack("<q4fffii", ptr, bx[0], bx[1], bx[2], bx[3], float(w), float(h), int(dst), 0)
    raw += struct.pack(f"<{B}i", *counts)
    table = _pack_ring.stage(raw, device)
    boxes = torch.empty(B, TM, 4, device=device, dtype=torch.float32)
    count = torch.empty(B, device=device, dtype=torch.int32)
    rc = lib.simvg_pack_targets(_p(table), len(rows), _p(boxes), _p(count), B, TM, _stream())
    _lib.check(rc, "simvg_pack_targets")
    return boxes, count


def match(logits, boxes, tboxes, tlabels, tcount, cost=(1.0, 5.0, 2.0)):
    """logits [L,B,nq,2], boxes [L,B,nq,4] fp32 -> int32 [L,B,nq] matched target index or -1."""
    lib = _lib.load()
    L, B, nq, _ = logits.shape
    TM = tboxes.shape[1]
    out = torch.empty(L, B, nq, device=logits.device, dtype=torch.int32)
    rc = lib.simvg_match(_p(logits), _p(boxes), _p(tboxes), _p(tlabels), _p(tcount), _p(out), L, B, nq, TM,
                         cost[0], cost[1], cost[2], _stream())
    _lib.check(rc, "simvg_match")
    return out


def postprocess(logits, boxes, wh, scale_factor=None):
    """logits [B,nq,C+1], boxes [B,nq,4] cxcywh (fp32), wh [B,4] = (w,h,w,h), scale_factor [B,4] | None ->
    (scores [B,nq], labels [B,nq] i64, xyxy [B,nq,4], keep [B,nq] bool, best_box [B,4], best_label [B] i64)."""
    lib = _lib.load()
    logits, boxes = logits.contiguous().float(), boxes.contiguous().float()
    B, nq, ncol = logits.shape
    dev = logits.device
    scores = torch.empty(B, nq, device=dev, dtype=torch.float32)
    labels = torch.empty(B, nq, device=dev, dtype=torch.int64)
    xyxy = torch.empty(B, nq, 4, device=dev, dtype=torch.float32)
    keep = torch.empty(B, nq, device=dev, dtype=torch.uint8)
    best_box = torch.empty(B, 4, device=dev, dtype=torch.float32)
    best_label = torch.empty(B, device=dev, dtype=torch.int64)
    sf = None if scale_factor is None else scale_factor.reshape(B, 4).contiguous().float()
    rc = lib.simvg_postprocess(_p(logits), _p(boxes), _p(wh.contiguous()), _p(sf), _p(scores), _p(labels), _p(xyxy), _p(keep),
                               _p(best_box), _p(best_label), B, nq, ncol, _stream())
    _lib.check(rc, "simvg_postprocess")
    return scores, labels, xyxy, keep.view(torch.bool), best_box, best_label


def soft_targets(logits, boxes, match_idx, tboxes, tcount):
    lib = _lib.load()
    B, nq, _ = logits.shape
    TM = tboxes.shape[1]
    dev = logits.device
    # the five zero-initialised outputs are slices of ONE fill (int32 zeros and fp32 zeros share the bit pattern)
    n_box, n_tm = B * TM * 4, B * TM
    buf = torch.zeros(n_box + 2 * n_tm + B + 4, device=dev)
    pboxes = buf[:n_box].view(B, TM, 4)
    pweight = buf[n_box:n_box + n_tm].view(B, TM)
    scal = buf[n_box + n_tm:n_box + n_tm + 4]
    plabels = buf[n_box + n_tm + 4:n_box + 2 * n_tm + 4].view(torch.int32).view(B, TM)
    pcount = buf[n_box + 2 * n_tm + 4:].view(torch.int32)
    rc = lib.simvg_soft_targets(_p(logits), _p(boxes), _p(match_idx), _p(tboxes), _p(tcount), _p(pboxes), _p(plabels),
                                _p(pcount), _p(pweight), _p(scal), B, nq, TM, _stream())
    _lib.check(rc, "simvg_soft_targets")
    return pboxes, plabels, pcount, pweight, scal


def criterion(logits, boxes, match_idx, tboxes, tlabels, num_boxes, wdist, coef_mode, coef, eos_coef=0.1,
              weights=(1.0, 5.0, 2.0)):
    """-> (out [1+3L] : total then per-layer class/bbox/giou, dlogits, dboxes)."""
    lib = _lib.load()
    L, B, nq, _ = logits.shape
    TM = tboxes.shape[1]
    # both gradients in one buffer: the backward's scaling by the upstream gradient is then one launch (Criterion.backward)
    nl, nb = logits.numel(), boxes.numel()
    grads = torch.empty(nl + nb, device=logits.device, dtype=logits.dtype)
    dlogits, dboxes = grads[:nl].view(logits.shape), grads[nl:].view(boxes.shape)
    out = torch.empty(1 + 3 * L, device=logits.device)
    rc = lib.simvg_criterion(_p(logits), _p(boxes), _p(match_idx), _p(tboxes), _p(tlabels), _p(num_boxes), _p(wdist),
                             _p(dlogits), _p(dboxes), _p(out), L, B, nq, TM, coef_mode, coef, eos_coef, weights[0],
                             weights[1], weights[2], _stream())
    _lib.check(rc, "simvg_criterion")
    return out, dlogits, dboxes


# ---------------------------------------------------------------------------------------------
# exact-fp32 forward mode
# ---------------------------------------------------------------------------------------------
def im2col_f32(img, P):
    lib = _lib.load()
    _chk(img, torch.float32, "img")
    B, Cc, S, _ = img.shape
    out = torch.empty(B * (S // P) ** 2, 3 * P * P, device=img.device, dtype=torch.float32)
    _lib.check(lib.simvg_im2col_f32(_p(img.contiguous()), _p(out), B, S, P, _stream()), "simvg_im2col_f32")
    return out


def attn_f32_fwd(qkv, B, H, Nv, Nt, pad=None):
    lib = _lib.load()
    _chk(qkv, torch.float32, "qkv")
    M, D3 = qkv.shape
    D = D3 // 3
    out = torch.empty(M, D, device=qkv.device, dtype=torch.float32)
    rc = lib.simvg_attn_f32_fwd(_p(qkv), qkv.stride(0), _p(out), out.stride(0), _p(pad), B, H, Nv, Nt, D,
                                (D // H) ** -0.5, _stream())
    _lib.check(rc, "simvg_attn_f32_fwd")
    return out


def linear_f32(x, W, b=None, out=None, act=0, accumulate=False):
    """out[M,N] (+)= x[M,K] W[N,K]^T + b (exact fp32 on MFMA f32); x, W may be strided 2-D views."""
    M, K = x.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    gemm_f32(x, x.stride(0), x.stride(1), W, W.stride(1), W.stride(0), out, M, N, K, bias=b, act=act, accumulate=accumulate)
    return out


_sumsq_ws = {}


def sumsq_accum(x, out):
    """out[0] += sum(x^2) over a contiguous fp32 tensor (numel % 4 == 0); deterministic (fixed-order partial sums)."""
    lib = _lib.load()
    _chk(x, torch.float32, "x")
    ws = _sumsq_ws.get(x.device)
    if ws is None:
        ws = _sumsq_ws[x.device] = torch.empty(2048, device=x.device, dtype=torch.float32)
    _lib.check(lib.simvg_sumsq(_p(x), x.numel(), _p(out), _p(ws), _stream()), "simvg_sumsq")
    return out


def adam_step(p, g, m, v, vmax, step_size, bc2_sqrt, beta1, beta2, eps, weight_decay=0.0, total_norm=None, max_norm=0.0):
    """fused clip-scale + Adam(amsgrad if vmax is given) over flat fp32 tensors, in place."""
    lib = _lib.load()
    t0 = _timer.start("adam") if _timer is not None else None
    rc = lib.simvg_adam_step(_p(p), _p(g), _p(m), _p(v), _p(vmax), p.numel(), step_size, bc2_sqrt, beta1, beta2, eps,
                             weight_decay, _p(total_norm), max_norm, _stream())
    if t0 is not None:
        _timer.stop("adam", t0, 0.0, 36.0 * p.numel())
    _lib.check(rc, "simvg_adam_step")


def resize_u8(src, full_hw, window=None):
    """src [H, W, 3] uint8 (HBM) -> window (y0, x0, h, w) of the image resized to full_hw = (h, w) with OpenCV's 8-bit
    INTER_LINEAR arithmetic; window=None: the whole resized image."""
    import ctypes
    lib = _lib.load()
    _chk(src, torch.uint8, "src")
    assert src.dim() == 3 and src.shape[2] == 3 and src.stride(2) == 1 and src.stride(1) == 3
    fh, fw = int(full_hw[0]), int(full_hw[1])
    y0, x0, oh, ow = (0, 0, fh, fw) if window is None else [int(v) for v in window]
    dst = torch.empty(oh, ow, 3, device=src.device, dtype=torch.uint8)
    rc = lib.simvg_resize_u8(_p(src), src.shape[0], src.shape[1], src.stride(0), _p(dst), dst.stride(0), oh, ow, fh, fw,
                             y0, x0, _stream())
    _lib.check(rc, "simvg_resize_u8")
    return dst


def normalize_pad_u8(src, mean, std, to_rgb, pad_hw, out=None):
    """src [h, w, 3] uint8 (HBM) -> fp32 [3, pad_h, pad_w]: optional BGR->RGB, (x - mean) * (1 / std), zero padding."""
    import ctypes
    lib = _lib.load()
    _chk(src, torch.uint8, "src")
    assert src.dim() == 3 and src.shape[2] == 3 and src.stride(2) == 1 and src.stride(1) == 3
    ph, pw = int(pad_hw[0]), int(pad_hw[1])
    if out is None:
        out = torch.empty(3, ph, pw, device=src.device, dtype=torch.float32)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    sd = (ctypes.c_float * 3)(*[float(v) for v in std])
    rc = lib.simvg_normalize_pad_u8(_p(src), src.stride(0), src.shape[0], src.shape[1], _p(out), ph, pw, m, sd, int(bool(to_rgb)),
                                    _stream())
    _lib.check(rc, "simvg_normalize_pad_u8")
    return out


class _ResizeJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src_h", C.c_int), ("src_w", C.c_int), ("src_row_bytes", C.c_long), ("dst", C.c_void_p),
                ("dst_row_bytes", C.c_long), ("out_h", C.c_int), ("out_w", C.c_int), ("full_h", C.c_int), ("full_w", C.c_int),
                ("win_y0", C.c_int), ("win_x0", C.c_int)]


class _FormatJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src_row_bytes", C.c_long), ("h", C.c_int), ("w", C.c_int), ("dst_chw", C.c_void_p),
                ("pad_h", C.c_int), ("pad_w", C.c_int)]


PREPROCESS_MAX_JOBS = 32


def resize_u8_batched(jobs):
    """jobs: list of (src [H,W,3] uint8 view, dst [oh,ow,3] uint8 view, full_hw, window | None) -- one launch per 32 jobs,
    frames of different geometry; same arithmetic as resize_u8."""
    lib = _lib.load()
    stream = _stream()
    for at in range(0, len(jobs), PREPROCESS_MAX_JOBS):
        chunk = jobs[at:at + PREPROCESS_MAX_JOBS]
        arr = (_ResizeJob * len(chunk))()
        for k, (src, dst, full_hw, window) in enumerate(chunk):
            fh, fw = int(full_hw[0]), int(full_hw[1])
            y0, x0, oh, ow = (0, 0, fh, fw) if window is None else window
            assert (int(dst.shape[0]), int(dst.shape[1])) == (oh, ow)
            arr[k] = _ResizeJob(src.data_ptr(), src.shape[0], src.shape[1], src.stride(0), dst.data_ptr(), dst.stride(0), oh, ow,
                                fh, fw, y0, x0)
        _lib.check(lib.simvg_resize_u8_batched(arr, len(chunk), stream), "simvg_resize_u8_batched")


def normalize_pad_u8_batched(jobs, mean, std, to_rgb):
    """jobs: list of (src [h,w,3] uint8 view, dst fp32 [3,ph,pw] view) sharing one normalisation"""
    lib = _lib.load()
    stream = _stream()
    m = (C.c_float * 3)(*[float(v) for v in mean])
    sd = (C.c_float * 3)(*[float(v) for v in std])
    for at in range(0, len(jobs), PREPROCESS_MAX_JOBS):
        chunk = jobs[at:at + PREPROCESS_MAX_JOBS]
        arr = (_FormatJob * len(chunk))()
        for k, (src, dst) in enumerate(chunk):
            arr[k] = _FormatJob(src.data_ptr(), src.stride(0), src.shape[0], src.shape[1], dst.data_ptr(), dst.shape[1], dst.shape[2])
        _lib.check(lib.simvg_normalize_pad_u8_batched(arr, len(chunk), m, sd, int(bool(to_rgb)), stream),
                   "simvg_normalize_pad_u8_batched")


def launch_resize_jobs(jobs):
    """jobs: numpy structured array laid out like simvg_resize_job[] (datasets.pipelines.RESIZE_JOB), addresses filled in"""
    lib = _lib.load()
    stream = _stream()
    jobs = numpy.ascontiguousarray(jobs)
    for at in range(0, len(jobs), PREPROCESS_MAX_JOBS):
        n = min(PREPROCESS_MAX_JOBS, len(jobs) - at)
        _lib.check(lib.simvg_resize_u8_batched(C.c_void_p(jobs.ctypes.data + at * jobs.itemsize), n, stream), "simvg_resize_u8_batched")


def launch_format_jobs(jobs, mean, std, to_rgb):
    lib = _lib.load()
    stream = _stream()
    jobs = numpy.ascontiguousarray(jobs)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    sd = (C.c_float * 3)(*[float(v) for v in std])
    for at in range(0, len(jobs), PREPROCESS_MAX_JOBS):
        n = min(PREPROCESS_MAX_JOBS, len(jobs) - at)
        _lib.check(lib.simvg_normalize_pad_u8_batched(C.c_void_p(jobs.ctypes.data + at * jobs.itemsize), n, m, sd, int(bool(to_rgb)), stream),
                   "simvg_normalize_pad_u8_batched")


def attn_f32_bwd(qkv, dout, B, H, Nv, Nt, pad=None):
    """exact-fp32 attention backward -> dqkv [M, 3D] fp32."""
    lib = _lib.load()
    _chk(qkv, torch.float32, "qkv"); _chk(dout, torch.float32, "dout")
    M, D3 = qkv.shape
    D = D3 // 3
    dqkv = torch.zeros(M, D3, device=qkv.device, dtype=torch.float32)
    rc = lib.simvg_attn_f32_bwd(_p(qkv), qkv.stride(0), _p(dout), dout.stride(0), _p(dqkv), dqkv.stride(0), _p(pad), B, H,
                                Nv, Nt, D, (D // H) ** -0.5, _stream())
    _lib.check(rc, "simvg_attn_f32_bwd")
    return dqkv


def gelu_f32(u, dy=None):
    """exact-erf GELU (dy None) or its backward dy * gelu'(u), elementwise fp32."""
    lib = _lib.load()
    _chk(u, torch.float32, "u")
    u = u.contiguous()
    out = torch.empty_like(u)
    _lib.check(lib.simvg_gelu_f32(_p(u), _p(dy.contiguous() if dy is not None else None), _p(out), u.numel(), _stream()), "simvg_gelu_f32")
    return out
