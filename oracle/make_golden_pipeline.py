"""Golden vectors for the image pipeline, produced by EXECUTING the reference's own
transform classes (LargeScaleJitter -> Resize -> Normalize -> Pad for training, Resize -> Normalize -> Pad for
evaluation; configs/**: img_scale (S, S), keep_ratio False, Pad size_divisor 32) on seeded synthetic uint8 images.

    python -m oracle.make_golden_pipeline        -> tests/golden/pipeline_golden.pt

The mmcv / OpenCV pixel leaves underneath are the restatements of oracle/pipeline_cpu.py (parity unpinned at that
boundary); composition, random crop search, box transforms and meta keys are the reference's.  TEST INFRASTRUCTURE ONLY."""
import os
import random

import numpy as np
import torch

from . import ref_loader

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pipeline_golden.pt")
MEAN, STD = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]

CASES = [  # (name, h, w, S, train, seed)
    ("val_landscape", 97, 131, 96, False, 1),
    ("val_portrait_odd", 150, 83, 160, False, 2),
    ("val_exact_half", 192, 192, 96, False, 3),          # exact 2x decimation: OpenCV's INTER_AREA fast path
    ("val_identity", 96, 96, 96, False, 4),
    ("train_a", 120, 160, 96, True, 5),
    ("train_b", 201, 133, 160, True, 6),
    ("train_c", 75, 230, 96, True, 7),
    ("train_d", 160, 160, 160, True, 8),
    ("train_e", 99, 140, 96, True, 9),
    ("train_f", 140, 99, 96, True, 10),
]


def make_input(h, w, seed):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = (xx * 3 + yy * 5) % 256
    img = np.stack([(base + 40 * c) % 256 for c in range(3)], -1).astype(np.int32)
    img = np.clip(img + rng.randint(-30, 31, size=img.shape), 0, 255).astype(np.uint8)     # structure + noise
    x0, y0 = rng.uniform(0, w * 0.5), rng.uniform(0, h * 0.5)
    bw, bh = rng.uniform(w * 0.2, w * 0.45), rng.uniform(h * 0.2, h * 0.45)
    return img, np.array([x0, y0, x0 + bw, y0 + bh], dtype=np.float64)


def run_reference(T, img, box, S, train, seed):
    random.seed(seed)
    np.random.seed(seed)
    results = dict(img=img.copy(), gt_bbox=box.copy(), ori_shape=img.shape, img_shape=img.shape, with_bbox=True, with_mask=False)
    steps = []
    if train:
        steps.append(T.LargeScaleJitter(out_max_size=S, jitter_min=0.3, jitter_max=1.4))
    steps += [T.Resize(img_scale=(S, S), keep_ratio=False), T.Normalize(mean=MEAN, std=STD), T.Pad(size_divisor=32)]
    for t in steps:
        results = t(results)
    return results


def main():
    T = ref_loader.load_pipelines()
    out = {}
    for name, h, w, S, train, seed in CASES:
        img, box = make_input(h, w, seed)
        r = run_reference(T, img, box, S, train, seed)
        out[name] = dict(h=h, w=w, S=S, train=train, seed=seed, img_in=torch.from_numpy(img), box_in=torch.from_numpy(box),
                         img=torch.from_numpy(np.ascontiguousarray(r["img"])), gt_bbox=torch.from_numpy(np.asarray(r["gt_bbox"], dtype=np.float64)),
                         img_shape=tuple(r["img_shape"]), pad_shape=tuple(r["pad_shape"]),
                         scale_factor=torch.from_numpy(np.asarray(r["scale_factor"], dtype=np.float64)), keep_ratio=bool(r["keep_ratio"]))
        print(name, "->", tuple(r["img"].shape), r["img_shape"], np.round(np.asarray(r["gt_bbox"]), 2), np.round(np.asarray(r["scale_factor"]), 4))
    torch.save(out, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
