"""Launch time of the streamed attention forward with parts switched off (library built with -DSIMVG_STREAM_ABLATE):
    SIMVG_EXTRA_FLAGS=-DSIMVG_STREAM_ABLATE SIMVG_LIB_SUFFIX=_abl python -m simvg_amd.build
    SIMVG_HIP_LIB=simvg_amd/lib/libsimvg_hip_abl.so python tools/dev/attn_stream_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
os.environ.setdefault("SIMVG_ATTN_STREAM", "1")
from simvg_amd import hip_ops as ops
dev = torch.device("cuda", 0)
B, H, N = int(os.environ.get("B", 64)), int(os.environ.get("H", 12)), 421
qkv = (torch.randn(B * N, 3 * H * 64) * 0.5).to(dev).to(ops.LP())
pad = torch.zeros(B, 20, dtype=torch.uint8); pad[:, 9:] = 1; pad = pad.to(dev)
out, lse = ops.attn_fwd(qkv, B, H, 401, 20, pad=pad)
NAMES = {1: "no exp2", 2: "no MFMA", 4: "no unit DMA / vmcnt", 8: "prio V", 16: "prio M", 32: "no fragment reads", 64: "no softmax",
         128: "no loop barriers", 256: "no epilogue"}
for abl in [int(x) for x in os.environ.get("ABLS", "0,1,2,4,32,64,128,256,66,98,102,358,486,511").split(",")]:
    os.environ["SIMVG_STREAM_ABL"] = str(abl)
    for _ in range(5):
        ops.attn_fwd(qkv, B, H, 401, 20, pad=pad, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.attn_fwd(qkv, B, H, 401, 20, pad=pad, out=out)
    e1.record(); e1.synchronize()
    print(f"abl {abl:4d}: {e0.elapsed_time(e1) * 20:.1f} us   [{', '.join(v for k, v in NAMES.items() if abl & k) or 'full kernel'}]", flush=True)
