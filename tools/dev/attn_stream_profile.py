"""Phase timeline of the streamed attention forward (csrc/attention_stream.hip built with -DSIMVG_STREAM_PROFILE into
libsimvg_hip_prof.so): workgroup 0's waves record s_memtime at the phase boundaries of every 64-key step.
    SIMVG_EXTRA_FLAGS=-DSIMVG_STREAM_PROFILE SIMVG_LIB_SUFFIX=_prof python -m simvg_amd.build      (dev container)
    SIMVG_HIP_LIB=simvg_amd/lib/libsimvg_hip_prof.so python tools/dev/attn_stream_profile.py       (GPU box)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops, _lib

dev = torch.device("cuda", 0)
B, H, N = int(os.environ.get("B", 64)), int(os.environ.get("H", 12)), 421
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = torch.zeros(8 * 64 * 8, dtype=torch.int64, device=dev)
lib.simvg_stream_profile_buffer(ctypes.c_void_p(buf.data_ptr()))
qkv = (torch.randn(B * N, 3 * H * 64) * 0.5).to(dev).to(ops.LP())
pad = torch.zeros(B, 20, dtype=torch.uint8); pad[:, 9:] = 1; pad = pad.to(dev)
for _ in range(3):
    out, lse = ops.attn_fwd(qkv, B, H, 401, 20, pad=pad)
torch.cuda.synchronize()
t = buf.cpu().view(8, 64, 8).double()
steps = 7 * ((B * H + 255) // 256)
t0 = t[:, 0, 0].min()
names = ["vmwait", "barrier", "issue", "K+QK", "V+softmax", "PV"]
print("s_memtime ticks (100 MHz constant clock: 1 tick = 10 ns); per wave: mean over steps of each phase")
for w in range(8):
    d = t[w, :steps, 1:7] - t[w, :steps, 0:6]
    tot = t[w, steps - 1, 6] - t[w, 0, 0]
    print(f"wave {w}: total {tot:.0f}  " + "  ".join(f"{n} {float(d[:, i].mean()):.1f}" for i, n in enumerate(names)))
w = 0
print("wave 0 per step:")
for s in range(steps):
    d = t[w, s, 1:7] - t[w, s, 0:6]
    print(f"  step {s:2d} start {t[w, s, 0] - t0:.0f}: " + " ".join(f"{float(x):.0f}" for x in d))
