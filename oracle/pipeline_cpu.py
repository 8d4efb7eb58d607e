"""CPU restatement of the image side of the reference's data pipeline: the mmcv / OpenCV
leaves that `simvg/datasets/pipelines/transforms.py` calls (`mmcv.imrescale`, `imresize`, `imnormalize`, `impad`,
`impad_to_multiple`, `rescale_size`).  TEST INFRASTRUCTURE ONLY.

mmcv and OpenCV are third-party dependencies of the reference that are absent from this image (and from
/root/reference), so the pixel arithmetic below restates their PUBLISHED algorithms and is **parity unpinned** against
the real libraries:
  * OpenCV 4.x `cv::resize(..., INTER_LINEAR)` on 8-bit images (imgproc/src/resize.cpp): half-pixel centres,
    fx = float((dx + 0.5) * scale - 0.5), taps clamped at the borders, 11-bit fixed-point coefficients
    (INTER_RESIZE_COEF_BITS = 11, cvRound = round-half-even), horizontal pass in int32, vertical pass
    ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2; an exact 2x2 decimation is routed to INTER_AREA;
  * mmcv 1.x `rescale_size` / `imrescale` (int(x * s + 0.5)), `imnormalize` (float32: (x - mean) * (1 / std) after an
    optional BGR->RGB swap), `impad` (bottom / right, constant).
What IS pinned (by executing the reference's own `LargeScaleJitter`, `Resize`, `Normalize`, `Pad` classes on top of
these leaves, `oracle/make_golden_pipeline.py`): the composition, the random crop search, every box transform and every
meta key."""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _taps(dst_n, src_n):
    """per output index: (s0, s1, a0, a1) with int16 coefficients, OpenCV resize.cpp INTER_LINEAR set-up"""
    scale = np.float64(src_n) / np.float64(dst_n)          # 1 / inv_scale
    d = np.arange(dst_n, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    f[lo] = 0.0
    s[lo] = 0
    hi = s >= src_n - 1
    f[hi] = 0.0
    s[hi] = src_n - 1
    a1 = np.rint(f * np.float32(COEF_SCALE)).astype(np.int32)
    a0 = np.rint((np.float32(1.0) - f) * np.float32(COEF_SCALE)).astype(np.int32)
    s1 = np.minimum(s + 1, src_n - 1)
    return s, s1, np.clip(a0, -32768, 32767), np.clip(a1, -32768, 32767)


def cv2_resize_linear_u8(src, dsize):
    """src [H, W, C] uint8, dsize = (w, h) -> [h, w, C] uint8"""
    src = np.ascontiguousarray(src)
    assert src.dtype == np.uint8 and src.ndim == 3
    sh, sw = src.shape[:2]
    dw, dh = int(dsize[0]), int(dsize[1])
    if (dw, dh) == (sw, sh):
        return src.copy()
    if sw == 2 * dw and sh == 2 * dh:                       # INTER_LINEAR with iscale 2x2 -> INTER_AREA fast path
        s = src.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, a0, a1 = _taps(dw, sw)
    y0, y1, b0, b1 = _taps(dh, sh)
    s = src.astype(np.int32)
    rows = s[:, x0, :] * a0[None, :, None] + s[:, x1, :] * a1[None, :, None]          # [sh, dw, C] horizontal pass
    r0, r1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def _scale_size(size, scale):
    w, h = size
    if isinstance(scale, (float, int)):
        scale = (scale, scale)
    return int(w * float(scale[0]) + 0.5), int(h * float(scale[1]) + 0.5)


def rescale_size(old_size, scale, return_scale=False):
    w, h = old_size
    if isinstance(scale, (float, int)):
        if scale <= 0:
            raise ValueError(f"Invalid scale {scale}, must be positive.")
        scale_factor = scale
    elif isinstance(scale, tuple):
        max_long_edge, max_short_edge = max(scale), min(scale)
        scale_factor = min(max_long_edge / max(h, w), max_short_edge / min(h, w))
    else:
        raise TypeError(f"Scale must be a number or tuple of int, but got {type(scale)}")
    new_size = _scale_size((w, h), scale_factor)
    return (new_size, scale_factor) if return_scale else new_size


def imresize(img, size, return_scale=False, interpolation="bilinear", out=None, backend=None):
    assert interpolation == "bilinear"
    h, w = img.shape[:2]
    resized = cv2_resize_linear_u8(img, size)
    if not return_scale:
        return resized
    return resized, size[0] / w, size[1] / h


def imrescale(img, scale, return_scale=False, interpolation="bilinear", backend=None):
    h, w = img.shape[:2]
    new_size, scale_factor = rescale_size((w, h), scale, return_scale=True)
    rescaled = imresize(img, new_size, interpolation=interpolation)
    return (rescaled, scale_factor) if return_scale else rescaled


def imnormalize(img, mean, std, to_rgb=True):
    img = img.astype(np.float32).copy()
    if to_rgb:
        img = img[..., ::-1].copy()
    mean = np.asarray(mean, dtype=np.float64).reshape(1, -1).astype(np.float32)
    stdinv = (1.0 / np.asarray(std, dtype=np.float64).reshape(1, -1)).astype(np.float32)
    return (img - mean) * stdinv


def impad(img, *, shape=None, padding=None, pad_val=0, padding_mode="constant"):
    assert shape is not None and padding_mode == "constant"
    h, w = img.shape[:2]
    out = np.full((shape[0], shape[1]) + img.shape[2:], pad_val, dtype=img.dtype)
    out[:h, :w] = img
    return out


def impad_to_multiple(img, divisor, pad_val=0):
    pad_h = int(np.ceil(img.shape[0] / divisor)) * divisor
    pad_w = int(np.ceil(img.shape[1] / divisor)) * divisor
    return impad(img, shape=(pad_h, pad_w), pad_val=pad_val)


def is_list_of(seq, expected_type):
    return isinstance(seq, list) and all(isinstance(x, expected_type) for x in seq)


def to_2tuple(x):
    return tuple(x) if isinstance(x, (list, tuple)) else (x, x)
