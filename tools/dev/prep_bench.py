"""Pre-processing kernels at the bench geometry: 64 frames 480x640 BGR u8 -> 640x640 -> normalised fp32 CHW."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simvg_amd import hip_ops as ops
B = 64
frames = [torch.randint(0, 256, (480, 640, 3), dtype=torch.uint8, device="cuda") for _ in range(B)]
out = torch.empty(B, 3, 640, 640, device="cuda")
mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
def batch():
    for i, f in enumerate(frames):
        r = ops.resize_u8(f, (640, 640))
        ops.normalize_pad_u8(r, mean, std, True, (640, 640), out=out[i])
for _ in range(5): batch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): batch()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
byts = B * (480 * 640 * 3 + 2 * 640 * 640 * 3 + 640 * 640 * 12)
print(f"preprocess batch of {B}: {ms:.3f} ms = {B / ms * 1e3:.0f} images/s, {byts / ms / 1e6:.0f} GB/s algorithmic")
big = torch.randint(0, 256, (3840, 5120, 3), dtype=torch.uint8, device="cuda")
for _ in range(3): r = ops.resize_u8(big, (5120, 5120))
torch.cuda.synchronize(); e0.record()
for _ in range(10): r = ops.resize_u8(big, (5120, 5120))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"resize_u8 3840x5120 -> 5120x5120: {ms * 1e3:.0f} us, {(big.numel() + r.numel()) / ms / 1e6:.0f} GB/s")
o = torch.empty(3, 5120, 5120, device="cuda")
for _ in range(3): ops.normalize_pad_u8(r, mean, std, True, (5120, 5120), out=o)
torch.cuda.synchronize(); e0.record()
for _ in range(10): ops.normalize_pad_u8(r, mean, std, True, (5120, 5120), out=o)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"normalize_pad 5120x5120: {ms * 1e3:.0f} us, {(r.numel() + o.numel() * 4) / ms / 1e6:.0f} GB/s")
