"""Device-side data pipeline: the reference's transform classes (`simvg/datasets/pipelines/transforms.py`,
`formatting.py`) with the same names, constructor arguments, result-dict keys, random-number consumption and box
arithmetic -- but `results["img"]` is a uint8 [H, W, 3] tensor in HBM and every pixel operation is a HIP kernel
(`csrc/preprocess.hip`).  SURVEY.md section 8 row f-3.

What is taken over verbatim in behaviour (pinned by `tests/golden/pipeline_golden.pt`, recorded from the reference's own
classes):  Resize (single scale, keep_ratio False/True), Normalize, Pad (size / size_divisor / pad_to_square),
LargeScaleJitter including its crop search, its "escape" branch (image and boxes left untouched, img_shape overwritten)
and its clipping of boxes to the crop.  LargeScaleJitter never materialises the rescaled image: rescale + crop is one
windowed resize.  `DeviceFormat` = Normalize + Pad + DefaultFormatBundle's HWC->CHW transpose in one kernel.

The pipeline's first step, `LoadImageAnnotationsFromFile` (file naming, decode, expression choice, token ids, boxes), is
`loading.py`; mask transforms are not built."""
import math
import random

import numpy
import torch

from .. import hip_ops as ops
from . import PIPELINES


def _scale_size(size, scale):
    w, h = size
    if isinstance(scale, (float, int)):
        scale = (scale, scale)
    return int(w * float(scale[0]) + 0.5), int(h * float(scale[1]) + 0.5)


def rescale_size(old_size, scale, return_scale=False):
    """mmcv.rescale_size: number -> that factor; (long, short) tuple -> largest factor that fits both edges."""
    w, h = old_size
    if isinstance(scale, (float, int)):
        if scale <= 0:
            raise ValueError(f"Invalid scale {scale}, must be positive.")
        scale_factor = scale
    elif isinstance(scale, tuple):
        scale_factor = min(max(scale) / max(h, w), min(scale) / min(h, w))
    else:
        raise TypeError(f"Scale must be a number or tuple of int, but got {type(scale)}")
    new_size = _scale_size((w, h), scale_factor)
    return (new_size, scale_factor) if return_scale else new_size


def _shape(img):
    return (int(img.shape[0]), int(img.shape[1]), int(img.shape[2]))


class DeferredFrame:
    """A uint8 [H, W, 3] frame in host memory plus the pixel operations still to be run on it.  In the two-stage loader
    (refsets.TwoStageLoader) the WHOLE pipeline runs in a DataLoader worker: every transform does its host work there --
    random draws, crop search, box arithmetic, shapes, meta keys -- and records its pixel operation here instead of
    launching it; the training process only moves the pinned frame to HBM and replays the recorded operations (two or
    three kernel launches per frame, no Python pipeline logic on the critical path)."""
    dtype = torch.uint8

    def __init__(self, pixels):
        self.pixels, self.shape, self.ops = pixels, tuple(int(s) for s in pixels.shape), []
        self.source_shape, self.offset = self.shape, None

    def resize(self, size_hw, window=None):
        self.ops.append(("resize", (int(size_hw[0]), int(size_hw[1])), None if window is None else tuple(int(v) for v in window)))
        h, w = (window[2], window[3]) if window is not None else size_hw
        self.shape = (int(h), int(w), self.shape[2])
        return self

    def format(self, mean, std, to_rgb, pad_hw):
        self.ops.append(("format", mean, std, bool(to_rgb), (int(pad_hw[0]), int(pad_hw[1]))))
        return self

    def pin_memory(self):
        if self.pixels is not None:
            self.pixels = self.pixels.pin_memory()
        return self

    def materialize(self, device, packed=None):
        """packed: the batch's frames as ONE uint8 tensor already in HBM (refsets.pack_host_batch); this frame is the
        slice [offset, offset + H0*W0*3) of it"""
        if packed is not None:
            h, w, c = self.source_shape
            img = packed[self.offset:self.offset + h * w * c].view(h, w, c)
        else:
            img = self.pixels.to(device, non_blocking=True)
        for op in self.ops:
            if op[0] == "resize":
                img = ops.resize_u8(img, op[1], op[2])
            else:
                img = ops.normalize_pad_u8(img, op[1], op[2], op[3], op[4])
        return img


def materialize_batch(frames, packed, device):
    """Replays the recorded pixel operations of a batch of DeferredFrames with BATCHED launches: round r executes op r of
    every frame that has one (resizes of all frames in one launch per 32, then the format passes).  `packed` = the batch's
    source frames as one uint8 tensor in HBM.  When every frame ends in a format pass of the same padded size the outputs
    are written straight into ONE [B, 3, H, W] tensor (no stacking copy afterwards).
    -> (list of per-frame tensors, stacked [B,3,H,W] tensor | None)"""
    cur = []
    for f in frames:
        h, w, c = f.source_shape
        cur.append(packed[f.offset:f.offset + h * w * c].view(h, w, c))
    rounds = max((len(f.ops) for f in frames), default=0)
    stacked = None
    for r in range(rounds):
        todo = [(i, f.ops[r]) for i, f in enumerate(frames) if r < len(f.ops)]
        resizes = [(i, op) for i, op in todo if op[0] == "resize"]
        formats = [(i, op) for i, op in todo if op[0] == "format"]
        if resizes:
            sizes = [((op[2][2], op[2][3]) if op[2] is not None else op[1]) for _, op in resizes]
            pool = torch.empty(sum(3 * h * w for h, w in sizes), device=device, dtype=torch.uint8)
            jobs, at = [], 0
            for (i, op), (h, w) in zip(resizes, sizes):
                dst = pool[at:at + 3 * h * w].view(h, w, 3)
                at += 3 * h * w
                jobs.append((cur[i], dst, op[1], op[2]))
            ops.resize_u8_batched(jobs)
            for (i, _), job in zip(resizes, jobs):
                cur[i] = job[1]
        if formats:
            pads = {op[4] for _, op in formats}
            same_cfg = all(numpy.array_equal(op[1], formats[0][1][1]) and numpy.array_equal(op[2], formats[0][1][2]) and op[3] == formats[0][1][3]
                           for _, op in formats)
            if len(formats) == len(frames) and len(pads) == 1 and same_cfg and all(r == len(f.ops) - 1 for f in frames):
                ph, pw = next(iter(pads))
                stacked = torch.empty(len(frames), 3, ph, pw, device=device, dtype=torch.float32)
                ops.normalize_pad_u8_batched([(cur[i], stacked[i]) for i, _ in formats], formats[0][1][1], formats[0][1][2], formats[0][1][3])
                for i, _ in formats:
                    cur[i] = stacked[i]
            else:
                for i, op in formats:
                    cur[i] = ops.normalize_pad_u8(cur[i], op[1], op[2], op[3], op[4])
    return cur, stacked


# ---- the batch's pixel work compiled in the WORKER into launch descriptors -------------------------------------------
# numpy mirrors of simvg_resize_job / simvg_format_job (include/simvg_hip.h); pointers hold byte OFFSETS until the consumer
# adds the base addresses of the buffers it allocated
RESIZE_JOB = numpy.dtype([("src", "<u8"), ("src_h", "<i4"), ("src_w", "<i4"), ("src_row_bytes", "<i8"), ("dst", "<u8"),
                          ("dst_row_bytes", "<i8"), ("out_h", "<i4"), ("out_w", "<i4"), ("full_h", "<i4"), ("full_w", "<i4"),
                          ("win_y0", "<i4"), ("win_x0", "<i4")], align=True)
FORMAT_JOB = numpy.dtype([("src", "<u8"), ("src_row_bytes", "<i8"), ("h", "<i4"), ("w", "<i4"), ("dst_chw", "<u8"),
                          ("pad_h", "<i4"), ("pad_w", "<i4")], align=True)
assert RESIZE_JOB.itemsize == 64 and FORMAT_JOB.itemsize == 40


def compile_batch_program(frames):
    """frames: DeferredFrames whose `offset` points into the batch's packed source buffer.  -> dict(rounds=[...],
    out_shape=(B, 3, ph, pw)) or None when the batch does not end in one common format pass (then the consumer replays the
    frames with `materialize_batch`).  Buffer ids: 0 = the packed source frames, k = the pool written by round k - 1."""
    if not frames or any(not f.ops or f.ops[-1][0] != "format" for f in frames):
        return None
    last = [f.ops[-1] for f in frames]
    if any(o[4] != last[0][4] or o[3] != last[0][3] or not numpy.array_equal(o[1], last[0][1]) or not numpy.array_equal(o[2], last[0][2])
           for o in last) or any(op[0] != "resize" for f in frames for op in f.ops[:-1]):
        return None
    B = len(frames)
    ph, pw = last[0][4]
    buf = [0] * B                                            # buffer holding each frame's current pixels
    off = [int(f.offset) for f in frames]
    shape = [f.source_shape[:2] for f in frames]
    rounds = []
    depth = max(len(f.ops) for f in frames) - 1              # resize rounds
    for r in range(depth):
        idx = [i for i, f in enumerate(frames) if r < len(f.ops) - 1]
        jobs = numpy.zeros(len(idx), dtype=RESIZE_JOB)
        src_buf = numpy.zeros(len(idx), dtype=numpy.int64)
        at = 0
        for k, i in enumerate(idx):
            _, (fh, fw), window = frames[i].ops[r]
            y0, x0, oh, ow = (0, 0, fh, fw) if window is None else window
            h, w = shape[i]
            jobs[k] = (off[i], h, w, 3 * w, at, 3 * ow, oh, ow, fh, fw, y0, x0)
            src_buf[k] = buf[i]
            buf[i], off[i], shape[i] = r + 1, at, (oh, ow)
            at += 3 * oh * ow
        rounds.append(dict(kind="resize", jobs=jobs, src_buf=src_buf, pool_bytes=at))
    jobs = numpy.zeros(B, dtype=FORMAT_JOB)
    src_buf = numpy.zeros(B, dtype=numpy.int64)
    for i in range(B):
        h, w = shape[i]
        jobs[i] = (off[i], 3 * w, h, w, i * 3 * ph * pw * 4, ph, pw)
        src_buf[i] = buf[i]
    rounds.append(dict(kind="format", jobs=jobs, src_buf=src_buf, mean=numpy.asarray(last[0][1], dtype=numpy.float32),
                       std=numpy.asarray(last[0][2], dtype=numpy.float32), to_rgb=bool(last[0][3])))
    return dict(rounds=rounds, out_shape=(B, 3, int(ph), int(pw)))


def run_batch_program(program, packed, device):
    """consumer side: allocate the pools, turn offsets into addresses (vectorised), launch -- no per-frame Python"""
    bases = [packed.data_ptr()]
    keep = [packed]
    out = torch.empty(program["out_shape"], device=device, dtype=torch.float32)
    for rd in program["rounds"]:
        jobs = rd["jobs"].copy()
        base = numpy.asarray(bases, dtype=numpy.uint64)
        jobs["src"] += base[rd["src_buf"]]
        if rd["kind"] == "resize":
            pool = torch.empty(max(int(rd["pool_bytes"]), 1), device=device, dtype=torch.uint8)
            keep.append(pool)
            bases.append(pool.data_ptr())
            jobs["dst"] += numpy.uint64(pool.data_ptr())
            ops.launch_resize_jobs(jobs)
        else:
            jobs["dst_chw"] += numpy.uint64(out.data_ptr())
            ops.launch_format_jobs(jobs, rd["mean"], rd["std"], rd["to_rgb"])
    return out


def tensorize(results):
    """ids / padding mask / boxes -> torch tensors (the tail of DefaultFormatBundle; the two-stage loader calls it in the
    training process)"""
    for key in ("ref_expr_inds", "text_attention_mask"):
        if key in results:
            results[key] = torch.as_tensor(numpy.asarray(results[key]))
    if "gt_bbox" in results and not torch.is_tensor(results["gt_bbox"]):
        results["gt_bbox"] = torch.as_tensor(numpy.asarray(results["gt_bbox"]))
    return results


def _resize(img, size_hw, window=None):
    return img.resize(size_hw, window) if isinstance(img, DeferredFrame) else ops.resize_u8(img, size_hw, window)


@PIPELINES.register_module()
class Resize:
    deferrable = True      # pixel work can be recorded on a DeferredFrame (two-stage loader)
    def __init__(self, img_scale=None, keep_ratio=True, interpolation="bilinear", backend="cv2"):
        if img_scale is None:
            self.img_scale = None
        else:
            self.img_scale = img_scale if isinstance(img_scale, list) else [img_scale]
            assert all(isinstance(s, tuple) for s in self.img_scale)
        if interpolation != "bilinear":
            raise NotImplementedError("only bilinear resizing is built (every reference config uses it)")
        self.keep_ratio = keep_ratio

    def __call__(self, results):
        scale = self.img_scale[0] if len(self.img_scale) == 1 else self.img_scale[numpy.random.randint(len(self.img_scale))]
        results["scale"] = scale
        img = results["img"]
        h, w = img.shape[:2]
        if self.keep_ratio:
            new_w, new_h = rescale_size((w, h), scale)
            img = _resize(img, (new_h, new_w))
            oh, ow = results["ori_shape"][:2]
            w_scale, h_scale = new_w / ow, new_h / oh
        else:
            img = _resize(img, (scale[1], scale[0]))
            w_scale, h_scale = scale[0] / w, scale[1] / h
        scale_factor = numpy.array([w_scale, h_scale, w_scale, h_scale], dtype=numpy.float32)
        results["img"] = img
        results["img_shape"] = _shape(img)
        results["pad_shape"] = _shape(img)
        results["scale_factor"] = scale_factor
        results["keep_ratio"] = self.keep_ratio
        if results["with_bbox"]:
            gt_bbox = results["gt_bbox"]
            results["gt_bbox"] = [b * scale_factor for b in gt_bbox] if isinstance(gt_bbox, list) else gt_bbox * scale_factor
        if results.get("with_mask"):
            raise NotImplementedError("mask transforms are outside this hot path")
        return results


@PIPELINES.register_module()
class Normalize:
    """Records the normalisation; the arithmetic is fused into `DeviceFormat` (one pass with padding and the CHW
    transpose).  Called on its own it produces the normalised fp32 HWC image like the reference."""
    deferrable = True      # pixel work can be recorded on a DeferredFrame (two-stage loader)

    def __init__(self, mean, std, to_rgb=True):
        self.mean = numpy.array(mean, dtype=numpy.float32)
        self.std = numpy.array(std, dtype=numpy.float32)
        self.to_rgb = to_rgb

    def __call__(self, results):
        results["img_norm_cfg"] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)
        results["_pending_normalize"] = True
        return results


@PIPELINES.register_module()
class Pad:
    deferrable = True      # pixel work can be recorded on a DeferredFrame (two-stage loader)
    def __init__(self, size=None, size_divisor=None, pad_to_square=False, pad_to_square_size=(640, 640), pad_val=0):
        self.size, self.size_divisor, self.pad_val = size, size_divisor, pad_val
        self.pad_to_square, self.pad_to_square_size = pad_to_square, pad_to_square_size
        if pad_to_square:
            assert size is None and size_divisor is None, "The size and size_divisor must be None when pad2square is True"
        else:
            assert size is not None or size_divisor is not None, "only one of size and size_divisor should be valid"
            assert size is None or size_divisor is None
        if pad_val != 0:
            raise NotImplementedError("only zero padding is built (every reference config uses it)")

    def __call__(self, results):
        if self.pad_to_square:
            self.size = self.pad_to_square_size
        h, w, c = _shape(results["img"])
        if self.size is not None:
            ph, pw = self.size
        else:
            ph = int(math.ceil(h / self.size_divisor)) * self.size_divisor
            pw = int(math.ceil(w / self.size_divisor)) * self.size_divisor
        results["pad_shape"] = (ph, pw, c)
        results["pad_fixed_size"] = self.size
        results["pad_size_divisor"] = self.size_divisor
        return results


@PIPELINES.register_module()
class LargeScaleJitter:
    deferrable = True      # pixel work can be recorded on a DeferredFrame (two-stage loader)
    def __init__(self, out_max_size=640, jitter_min=0.3, jitter_max=1.4, min_iou_thr=0.3, crop_iou_thr=[0.5, 0.6, 0.7, 0.8, 0.9]):
        self.out_max_size, self.jitter_min, self.jitter_max = out_max_size, jitter_min, jitter_max
        self.crop_iou_thr, self.min_iou_thr, self.jitter_times = crop_iou_thr, min_iou_thr, 100

    @staticmethod
    def _bbox_overlaps(crop_bbox, gt_bbox):
        lt = numpy.maximum(crop_bbox[:2], gt_bbox[:2])
        rb = numpy.minimum(crop_bbox[2:], gt_bbox[2:])
        wh = rb - lt
        return wh[0] * wh[1] / ((gt_bbox[2] - gt_bbox[0]) * (gt_bbox[3] - gt_bbox[1]))

    def _search_window(self, span_w, span_h, w_out, h_out, gt_bbox):
        """Random w_out x h_out windows inside the rescaled image, judged by the share of the target box they contain:
        thresholds from strict to lenient, `jitter_times` draws each, the first window that reaches the current threshold
        wins.  -> (window xyxy | None, coverage of the best window seen, that window, last offset drawn).  Consumes two
        `random.random()` per draw, like the reference's loop (transforms.py:262-286)."""
        windows, coverage, hit = [], [], None
        for thr in reversed(self.crop_iou_thr):
            for _ in range(self.jitter_times):
                ox, oy = random.random() * span_w, random.random() * span_h
                win = numpy.array([ox, oy, ox + w_out, oy + h_out])
                cov = self._bbox_overlaps(win, gt_bbox) if gt_bbox is not None else 0.0
                windows.append(win)
                coverage.append(cov)
                if cov >= thr:
                    hit = win
                    break
            if hit is not None:
                break
        top = max(coverage)
        # the reference remembers the FIRST draw that improved on everything before it (strict >, starting from 0), and
        # falls back to index -1 when no window touched the box at all
        runner_up = windows[coverage.index(top)] if top > 0 else windows[-1]
        return hit, top, runner_up, (ox, oy)

    def __call__(self, results):
        if results.get("with_mask"):
            raise NotImplementedError("mask transforms are outside this hot path")
        img = results["img"]
        h, w = results["ori_shape"][:2]
        channels = int(img.shape[2])
        jitter = self.jitter_min + random.random() * (self.jitter_max - self.jitter_min)
        new_w, new_h = rescale_size((int(img.shape[1]), int(img.shape[0])), jitter * (self.out_max_size / max(h, w)))
        boxes = None
        if results["with_bbox"]:
            grow = numpy.array([new_w / w, new_h / h, new_w / w, new_h / h])
            raw = results["gt_bbox"]
            boxes = [b * grow for b in raw] if isinstance(raw, list) else raw * grow
        out_w, out_h, window = new_w, new_h, None              # window = (y0, x0, h, w) inside the rescaled image
        if jitter > 1.0:                                       # larger than the canvas: cut an out_max_size window out of it
            w_out, h_out = rescale_size((w, h), (self.out_max_size, self.out_max_size))
            hit, top, runner_up, last_offset = self._search_window(new_w - w_out, new_h - h_out, w_out, h_out, boxes)
            if hit is None and top < self.min_iou_thr:         # no acceptable window: the sample passes through untouched
                results.update(img_shape=(new_h, new_w, channels), pad_shape=(new_h, new_w, channels),
                               scale_factor=numpy.array([1.0, 1.0, 1.0, 1.0]), keep_ratio=True)
                return results
            x0, y0, x1, y1 = [int(v) for v in (hit if hit is not None else runner_up).astype(numpy.uint32)]
            window, out_w, out_h = (y0, x0, y1 - y0, x1 - x0), x1 - x0, y1 - y0
            assert (out_h, out_w) == (h_out, w_out)
            if boxes is not None:      # shifted by the LAST offset drawn, also when an earlier window is used (reference quirk)
                boxes = boxes - numpy.array([last_offset[0], last_offset[1], last_offset[0], last_offset[1]])
        if boxes is not None:
            boxes[0::2] = numpy.clip(boxes[0::2], 0, out_w - 1)
            boxes[1::2] = numpy.clip(boxes[1::2], 0, out_h - 1)
            assert boxes[0] >= 0 and boxes[1] >= 0 and boxes[2] <= out_w and boxes[3] <= out_h
            results["gt_bbox"] = boxes
        img = _resize(img, (new_h, new_w), window)             # rescale (+ crop) in one pass over the source image
        results.update(img=img, img_shape=_shape(img), pad_shape=_shape(img), keep_ratio=True,
                       scale_factor=numpy.array([out_w / w, out_h / h, out_w / w, out_h / h]))
        return results


@PIPELINES.register_module()
class DefaultFormatBundle:
    """formatting.py:18-104: image -> fp32 CHW tensor (here: normalisation + padding + transpose in ONE kernel, straight
    from the uint8 image), ids / mask / boxes -> tensors."""
    deferrable = True      # pixel work can be recorded on a DeferredFrame (two-stage loader)

    def __call__(self, results):
        img = results["img"]
        if img.dtype != torch.uint8:
            raise TypeError("DefaultFormatBundle (device) expects the uint8 HWC image; Normalize / Pad only record their settings")
        cfg = results.get("img_norm_cfg") if results.pop("_pending_normalize", False) else None
        ph, pw = results.get("pad_shape", _shape(img))[:2]
        results.setdefault("pad_shape", _shape(img))
        results.setdefault("scale_factor", 1.0)
        if cfg is None:
            cfg = dict(mean=numpy.zeros(3, dtype=numpy.float32), std=numpy.ones(3, dtype=numpy.float32), to_rgb=False)
            results.setdefault("img_norm_cfg", cfg)
        if isinstance(img, DeferredFrame):
            results["img"] = img.format(cfg["mean"], cfg["std"], cfg["to_rgb"], (ph, pw))
        else:
            results["img"] = ops.normalize_pad_u8(img, cfg["mean"], cfg["std"], cfg["to_rgb"], (ph, pw))
        if isinstance(img, DeferredFrame):
            # worker process: ids / mask / boxes stay numpy (pickled inline); as torch tensors each of them would travel to the
            # training process through its own shared-memory file descriptor (~0.2 ms apiece: 12 ms per batch of 64)
            for key in ("ref_expr_inds", "text_attention_mask"):
                if key in results:
                    results[key] = numpy.asarray(results[key])
            if results.get("with_bbox"):
                results["gt_bbox"] = numpy.asarray(results["gt_bbox"])
            return results
        tensorize(results)
        return results


@PIPELINES.register_module()
class CollectData:
    deferrable = True      # pixel work can be recorded on a DeferredFrame (two-stage loader)
    def __init__(self, keys, meta_keys=("filename", "expression", "ori_shape", "img_shape", "pad_shape", "scale_factor")):
        self.keys, self.meta_keys = keys, meta_keys

    def __call__(self, results):
        data = {k: results[k] for k in self.keys}
        data["img_metas"] = {k: results[k] for k in self.meta_keys if k in results}
        return data


class Compose:
    def __init__(self, transforms):
        self.transforms = [PIPELINES.build(t) if isinstance(t, dict) else t for t in transforms]

    def __call__(self, results):
        for t in self.transforms:
            results = t(results)
            if results is None:
                return None
        return results
