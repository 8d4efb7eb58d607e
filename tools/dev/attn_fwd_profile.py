"""Phase timeline of the resident attention forward (development build: SIMVG_EXTRA_FLAGS="-DFWD_PROFILE"
SIMVG_LIB_SUFFIX=_fwdprof python -m simvg_amd.build; run with SIMVG_HIP_LIB=simvg_amd/lib/libsimvg_hip_fwdprof.so).
Workgroups 0 (first residency round) and 600 (third) stamp s_memtime per wave: entry, loads issued, LDS written, barrier passed,
then per query strip: end of the QK^T phase (row maximum known), end of the strip (stores issued)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops, _lib
B, H, Nv, Nt, d = 64, 12, 401, 20, 64
N, D = Nv + Nt, H * d
qkv = torch.randn(B * N, 3 * D, device="cuda").to(ops.LP())
lib = _lib.load()
buf = torch.zeros(2 * 12 * 16, device="cuda", dtype=torch.float32)
lib.simvg_attn_fwd_profile_buf.argtypes = [C.c_void_p]
lib.simvg_attn_fwd_profile_buf(C.c_void_p(buf.data_ptr()))
for _ in range(4):
    out, lse = ops.attn_fwd(qkv, B, H, Nv, Nt)
torch.cuda.synchronize()
t = buf.view(torch.int32).cpu().view(2, 12, 16).long() & 0xffffffff
for wg in range(2):
    t0 = int(t[wg, :, 0].min())
    print(f"workgroup {'0' if wg == 0 else '600'} (cycles from the first wave's entry)")
    for w in range(12):
        r = [(int(x) - t0) & 0xffffffff for x in t[wg, w]]
        strips = [(r[4 + 2 * i], r[5 + 2 * i]) for i in range(3) if t[wg, w, 5 + 2 * i] != 0]
        print(f"  wave {w:2d}: entry {r[0]:6d} issued {r[1]:6d} written {r[2]:6d} barrier {r[3]:6d} | " +
              " | ".join(f"qk {a:6d} end {b:6d}" for a, b in strips))
