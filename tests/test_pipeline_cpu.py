"""CPU: host logic of the device-side pipeline classes (random crop search, box arithmetic, meta keys, composition)
against the golden run of the REFERENCE's own transform classes (`oracle/make_golden_pipeline.py`).  The two HIP pixel
kernels are replaced here by their CPU restatement (oracle/pipeline_cpu.py); `tests/test_pipeline_gpu.py` checks the
kernels themselves bit-exactly against that restatement and the whole pipeline on the GPU."""
import os
import random

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = torch.load(os.path.join(HERE, "golden", "pipeline_golden.pt"), weights_only=False)
MEAN, STD = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]


def oracle_resize(src, full_hw, window=None):
    from oracle import pipeline_cpu as P
    full = P.cv2_resize_linear_u8(src.cpu().numpy(), (int(full_hw[1]), int(full_hw[0])))
    if window is not None:
        y0, x0, h, w = window
        full = full[y0:y0 + h, x0:x0 + w]
    return torch.from_numpy(np.ascontiguousarray(full))


def oracle_normalize_pad(src, mean, std, to_rgb, pad_hw, out=None):
    from oracle import pipeline_cpu as P
    x = P.imnormalize(src.cpu().numpy(), mean, std, to_rgb)
    x = P.impad(x, shape=(int(pad_hw[0]), int(pad_hw[1])))
    return torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))


def run_device_pipeline(case, device="cpu"):
    from simvg_amd.datasets.pipelines import Compose
    g = GOLD[case]
    S = g["S"]
    steps = []
    if g["train"]:
        steps.append(dict(type="LargeScaleJitter", out_max_size=S, jitter_min=0.3, jitter_max=1.4))
    steps += [dict(type="Resize", img_scale=(S, S), keep_ratio=False), dict(type="Normalize", mean=MEAN, std=STD),
              dict(type="Pad", size_divisor=32), dict(type="DefaultFormatBundle"),
              dict(type="CollectData", keys=["img", "gt_bbox"], meta_keys=("ori_shape", "img_shape", "pad_shape", "scale_factor", "keep_ratio"))]
    random.seed(g["seed"])
    np.random.seed(g["seed"])
    img = g["img_in"].to(device)
    results = dict(img=img, gt_bbox=g["box_in"].numpy().copy(), ori_shape=tuple(img.shape), img_shape=tuple(img.shape),
                   with_bbox=True, with_mask=False)
    return Compose(steps)(results)


def check_against_golden(case, out, atol):
    g = GOLD[case]
    ref_img = g["img"].permute(2, 0, 1)
    assert tuple(out["img"].shape) == tuple(ref_img.shape) and out["img"].dtype == torch.float32
    assert float((out["img"].cpu() - ref_img).abs().max()) <= atol
    assert torch.allclose(out["gt_bbox"].double().cpu(), g["gt_bbox"], rtol=0, atol=1e-9)
    m = out["img_metas"]
    assert tuple(m["img_shape"]) == g["img_shape"] and tuple(m["pad_shape"]) == g["pad_shape"]
    assert np.allclose(np.asarray(m["scale_factor"], dtype=np.float64), g["scale_factor"].numpy(), rtol=0, atol=1e-12)
    assert bool(m["keep_ratio"]) == g["keep_ratio"]


@pytest.mark.parametrize("case", sorted(GOLD))
def test_pipeline_host_logic_matches_reference(case, monkeypatch):
    from simvg_amd import hip_ops
    monkeypatch.setattr(hip_ops, "resize_u8", oracle_resize)
    monkeypatch.setattr(hip_ops, "normalize_pad_u8", oracle_normalize_pad)
    out = run_device_pipeline(case)
    check_against_golden(case, out, atol=0.0)


def test_restated_cv2_resize_properties():
    """sanity of the restatement itself: identity, constant images, exact 2x decimation, monotone ramps"""
    from oracle import pipeline_cpu as P
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=(37, 53, 3)).astype(np.uint8)
    assert np.array_equal(P.cv2_resize_linear_u8(img, (53, 37)), img)
    const = np.full((20, 30, 3), 173, np.uint8)
    assert np.array_equal(P.cv2_resize_linear_u8(const, (77, 41)), np.full((41, 77, 3), 173, np.uint8))
    big = rng.randint(0, 256, size=(40, 60, 3)).astype(np.int32)
    half = P.cv2_resize_linear_u8(big.astype(np.uint8), (30, 20))
    assert np.array_equal(half, ((big[0::2, 0::2] + big[0::2, 1::2] + big[1::2, 0::2] + big[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    ramp = np.repeat(np.arange(0, 250, 5, dtype=np.uint8)[None, :, None], 3, 2).repeat(8, 0)     # [8, 50, 3]
    up = P.cv2_resize_linear_u8(ramp, (125, 8)).astype(np.int32)
    assert (np.diff(up[0, :, 0]) >= 0).all() and up.min() == 0 and up.max() == 245
    assert P.rescale_size((640, 480), 0.5) == (320, 240) and P.rescale_size((500, 375), (640, 640)) == (640, 480)
