"""Streamed persistent attention forward (csrc/attention_stream.hip) against the resident kernel and an fp32 PyTorch
reference on the GPU, at training-size launches; then isolated timings of both."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
os.environ.setdefault("SIMVG_ATTN_STREAM", "1")
from simvg_amd import hip_ops as ops

dev = torch.device("cuda", 0)


def reference(qkv, pad, B, H, Nv, Nt):
    N, D = Nv + Nt, H * 64
    x = qkv.float()
    tok = torch.cat([x[:B * Nv].view(B, Nv, 3 * D), x[B * Nv:].view(B, Nt, 3 * D)], 1)
    q, k, v = tok.split(D, -1)
    q = q.view(B, N, H, 64).transpose(1, 2) * 64 ** -0.5
    k = k.view(B, N, H, 64).transpose(1, 2)
    v = v.view(B, N, H, 64).transpose(1, 2)
    w = q @ k.transpose(-1, -2)
    if pad is not None:
        kpm = torch.cat([torch.zeros(B, Nv, dtype=torch.bool, device=dev), pad.bool()], 1)
        w = w.masked_fill(kpm[:, None, None, :], float("-inf"))
    o = (torch.softmax(w, -1) @ v).transpose(1, 2).reshape(B, N, D)
    o = torch.cat([o[:, :Nv].reshape(B * Nv, D), o[:, Nv:].reshape(B * Nt, D)], 0)
    return o, torch.logsumexp(w, -1).reshape(B * H, N)


def run(B, H, Nv=401, Nt=20, use_pad=True, scale=1.0):
    N, D = Nv + Nt, H * 64
    g = torch.Generator(device="cpu").manual_seed(B * 131 + H)
    qkv = (torch.randn(B * N, 3 * D, generator=g) * scale).to(dev).to(ops.LP())
    pad = None
    if use_pad:
        pad = torch.zeros(B, Nt, dtype=torch.uint8)
        for b in range(B):
            pad[b, Nt - (3 + 5 * b) % Nt:] = 1
        pad = pad.to(dev)
    os.environ.pop("SIMVG_ATTN_RESIDENT", None)
    out_s, lse_s = ops.attn_fwd(qkv, B, H, Nv, Nt, pad=pad)
    torch.cuda.synchronize()
    os.environ["SIMVG_ATTN_RESIDENT"] = "1"
    out_r, lse_r = ops.attn_fwd(qkv, B, H, Nv, Nt, pad=pad)
    torch.cuda.synchronize()
    os.environ.pop("SIMVG_ATTN_RESIDENT", None)
    o_ref, lse_ref = reference(qkv, pad, B, H, Nv, Nt)
    e_s = float((out_s.float() - o_ref).abs().max()); e_r = float((out_r.float() - o_ref).abs().max())
    l_s = float((lse_s - lse_ref).abs().max()); l_r = float((lse_r - lse_ref).abs().max())
    d = float((out_s.float() - out_r.float()).abs().max())
    print(f"B={B} H={H} pad={use_pad} scale={scale}: |out-ref| stream {e_s:.3e} resident {e_r:.3e}; |lse-ref| {l_s:.3e} / {l_r:.3e}; "
          f"stream-resident {d:.3e}; nan {bool(torch.isnan(out_s.float()).any())}", flush=True)
    return e_s, e_r, l_s


def timed(fn, reps=100):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


if __name__ == "__main__":
    for B, H in [(22, 12), (43, 12), (64, 12), (17, 16), (32, 16)]:
        run(B, H)
    run(64, 12, use_pad=False)
    run(24, 12, scale=4.0)
    for B, H in [(64, 12), (32, 16), (64, 16)]:
        N, D = 421, H * 64
        qkv = (torch.randn(B * N, 3 * D) * 0.5).to(dev).to(ops.LP())
        pad = torch.zeros(B, 20, dtype=torch.uint8); pad[:, 9:] = 1; pad = pad.to(dev)
        out, lse = ops.attn_fwd(qkv, B, H, 401, 20, pad=pad)
        os.environ.pop("SIMVG_ATTN_RESIDENT", None)
        ts = timed(lambda: ops.attn_fwd(qkv, B, H, 401, 20, pad=pad, out=out))
        os.environ["SIMVG_ATTN_RESIDENT"] = "1"
        tr = timed(lambda: ops.attn_fwd(qkv, B, H, 401, 20, pad=pad, out=out))
        os.environ.pop("SIMVG_ATTN_RESIDENT", None)
        fl = 4.0 * B * H * N * N * 64
        print(f"B={B} H={H}: stream {ts:.1f} us ({fl / ts / 1e6:.0f} TF/s), resident {tr:.1f} us ({fl / tr / 1e6:.0f} TF/s)", flush=True)
