"""Phase timeline of the streamed attention forward (csrc/attention_stream.hip built with -DSIMVG_STREAM_PROFILE into
libsimvg_hip_prof.so): workgroup 0's waves record s_memtime at the phase boundaries of every 64-key step.
    SIMVG_EXTRA_FLAGS=-DSIMVG_STREAM_PROFILE SIMVG_LIB_SUFFIX=_prof python -m simvg_amd.build      (dev container)
    SIMVG_HIP_LIB=simvg_amd/lib/libsimvg_hip_prof.so python tools/dev/attn_stream_profile.py       (GPU box)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
os.environ.setdefault("SIMVG_ATTN_STREAM", "1")
from simvg_amd import hip_ops as ops, _lib

dev = torch.device("cuda", 0)
B, H, N = int(os.environ.get("B", 64)), int(os.environ.get("H", 12)), 421
lib = ctypes.CDLL(_lib.LIB_PATH)
SLOTS, STEPS = 4, 64
buf = torch.zeros(8 * STEPS * SLOTS, dtype=torch.int64, device=dev)
lib.simvg_stream_profile_buffer(ctypes.c_void_p(buf.data_ptr()))
qkv = (torch.randn(B * N, 3 * H * 64) * 0.5).to(dev).to(ops.LP())
pad = torch.zeros(B, 20, dtype=torch.uint8); pad[:, 9:] = 1; pad = pad.to(dev)
units = min(STEPS, 14 * ((B * H + 255) // 256))
print("s_memtime = shader cycles; per wave, mean over units 2..11: wait at the M barrier | M phase | wait at the V barrier | V phase")
for abl in [int(x) for x in os.environ.get("ABLS", "0").split(",")]:
    os.environ["SIMVG_STREAM_ABL"] = str(abl)
    for _ in range(3):
        out, lse = ops.attn_fwd(qkv, B, H, 401, 20, pad=pad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.attn_fwd(qkv, B, H, 401, 20, pad=pad, out=out)
    e1.record(); e1.synchronize()
    t = buf.cpu().view(8, STEPS, SLOTS).double()
    print(f"--- ablation {abl}: {e0.elapsed_time(e1) * 50:.1f} us per launch")
    for w in (0, 3, 4, 7):
        bm = t[w, 2:12, 1] - t[w, 2:12, 0]
        mp = t[w, 2:12, 2] - t[w, 2:12, 1]
        bv = t[w, 2:12, 3] - t[w, 2:12, 2]
        vp = t[w, 3:13, 0] - t[w, 2:12, 3]
        tot = t[w, units - 1, 3] - t[w, 0, 0]
        print(f"  wave {w}: total {tot:.0f}  M-barrier {float(bm.mean()):.0f}  M {float(mp.mean()):.0f}  V-barrier {float(bv.mean()):.0f}  V {float(vp.mean()):.0f}"
              f"   head boundary (units 12..15): {t[w, 16, 0] - t[w, 12, 0]:.0f}")
