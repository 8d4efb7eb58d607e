"""Isolated timing of the encoder self-attention kernels (forward, backward) on the bench geometry."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

B = int(os.environ.get("B", 64))
H = int(os.environ.get("H", 12))
r = bench.attention_roofline(B, H, 401, 20, 64, torch.device("cuda", 0), reps=100)
print({k: round(v, 2) if isinstance(v, float) else v for k, v in r.items()})
