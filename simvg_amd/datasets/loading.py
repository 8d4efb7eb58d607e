"""File-bound head of the data pipeline: `LoadImageAnnotationsFromFile` with the reference's
name, constructor arguments and result keys (`simvg/datasets/pipelines/loading.py:22-279`), feeding the DEVICE-side
transforms of `pipelines.py`:

  annotation record -> image file name (per-dataset naming rules, :80-96) -> decode -> uint8 [H, W, 3] BGR tensor in HBM
                    -> one expression (`numpy.random.choice`, same draw as the reference) -> clean_string -> token ids
                       (XLM-R sentencepiece ids + padding mask for use_token_type="beit3", word ids for "default")
                    -> gt box(es) xywh -> xyxy, clipped to the image (GRefCOCO: the boxes and `target` records of the
                       chosen expression).

Decode is PIL on the host (no JPEG engine is exposed on the device side here); the uint8 frame is what crosses PCIe
(a 640x480 frame is 0.9 MB against 4.9 MB for the normalised fp32 tensor the reference's workers hand over).
Not built: `with_mask` (pycocotools RLE / polygon rasterisation), use_token_type "bert" / "copus"."""
import copy
import os.path as osp
import re

import numpy
import torch

from . import PIPELINES

_DATASETS = ("GRefCOCO", "RefCOCOUNC", "RefCOCOGoogle", "RefCOCOgUMD", "RefCOCOgGoogle", "RefCOCOPlusUNC",
             "ReferItGameBerkeley", "Flickr30k", "Mixed")
_PUNCT = re.compile(r"([.,'!?\"()*#:;])")


def clean_string(expression):
    """lower-case, drop . , ' ! ? " ( ) * # : ; and turn - and / into blanks (loading.py:14-19)"""
    return _PUNCT.sub("", expression.lower()).replace("-", " ").replace("/", " ")


def image_path(dataset, imgsfile, ann):
    if "ReferItGame" in dataset or "Flickr30k" in dataset:
        return osp.join(imgsfile, "%d.jpg" % ann["image_id"])
    if "RefCOCO" in dataset:
        return osp.join(imgsfile, "COCO_train2014_%012d.jpg" % ann["image_id"])
    if dataset == "Mixed":
        source = ann["data_source"]
        name = ("COCO_train2014_%012d.jpg" if "coco" in source else "%d.jpg") % ann["image_id"]
        return osp.join(imgsfile[source], name)
    raise ValueError(f"no image naming rule for dataset {dataset!r}")


def decode_image(path, color_type="color"):
    """-> uint8 [H, W, 3] in BGR order (mmcv.imfrombytes' default channel order, which Normalize(to_rgb=True) undoes).
    The EXIF orientation is applied, as cv2.imdecode(IMREAD_COLOR) behind mmcv.imfrombytes(flag='color') does (a few
    COCO train2014 / Flickr30k JPEGs carry the tag: without it h / w, the box clipping and the pixels would differ)."""
    from PIL import Image, ImageOps
    if color_type != "color":
        raise NotImplementedError("only color_type='color' is built")
    with Image.open(path) as im:
        rgb = numpy.asarray(ImageOps.exif_transpose(im).convert("RGB"))
    return numpy.ascontiguousarray(rgb[:, :, ::-1])


def _xyxy_clipped(box_xywh, h, w):
    b = numpy.array(box_xywh, dtype=numpy.float64)
    b[2] += b[0]
    b[3] += b[1]
    b[0::2] = numpy.clip(b[0::2], 0, w - 1)
    b[1::2] = numpy.clip(b[1::2], 0, h - 1)
    return b


@PIPELINES.register_module()
class LoadImageAnnotationsFromFile:
    host_side = True          # runs in DataLoader workers when the loader is two-stage (refsets.TwoStageLoader)

    def __init__(self, dataset="RefCOCOUNC", color_type="color", backend=None, file_client_cfg=dict(backend="disk"),
                 max_token=15, with_bbox=False, with_mask=False, use_token_type="default",
                 spm_path="pretrain_weights/beit3.spm", device=None):
        assert with_bbox or with_mask
        assert dataset in _DATASETS
        if with_mask:
            raise NotImplementedError("with_mask=True needs the pycocotools mask codec: segmentation targets are not built")
        if use_token_type not in ("default", "beit3"):
            raise NotImplementedError(f"use_token_type={use_token_type!r}: only 'default' (word ids) and 'beit3' (XLM-R "
                                      "sentencepiece ids) are built")
        if file_client_cfg.get("backend", "disk") != "disk":
            raise NotImplementedError("only the 'disk' file backend is built")
        self.dataset, self.color_type, self.backend = dataset, color_type, backend
        self.max_token, self.with_bbox, self.with_mask = max_token, with_bbox, with_mask
        self.use_token_type, self.spm_path = use_token_type, spm_path
        self.device = device
        self._tokenizer = None
        self.random_ind = 0

    @property
    def tokenizer(self):
        if self._tokenizer is None:
            from .tokenizer import XLMRTokenizer
            self._tokenizer = XLMRTokenizer(self.spm_path)
        return self._tokenizer

    def _device(self):
        if self.device is not None:
            return torch.device(self.device)
        return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

    def __call__(self, results, host_only=False):
        """host_only: leave the decoded frame in (CPU) memory -- the consumer moves it to the GPU"""
        ann = results["ann"]
        path = image_path(self.dataset, results["imgsfile"], ann)
        frame = decode_image(path, self.color_type)
        shape = tuple(int(s) for s in frame.shape)
        img = torch.from_numpy(frame)
        results.update(filename=path, img=img if host_only else img.to(self._device(), non_blocking=True), img_shape=shape,
                       ori_shape=shape)
        # ---- expression: one of the record's expressions, drawn like the reference draws it
        expressions = ann["expressions"]
        self.random_ind = int(numpy.random.choice(list(range(len(expressions)))))
        expression = clean_string(expressions[self.random_ind])
        if self.use_token_type == "beit3":
            ids, mask = self.tokenizer.encode_expression(expression, self.max_token)
            results["ref_expr_inds"] = numpy.array(ids, dtype=int)
            results["text_attention_mask"] = numpy.array(mask, dtype=int)
        else:
            table = results["token2idx"]
            ids = torch.zeros(self.max_token, dtype=torch.long)
            for i, word in enumerate(expression.split()[:self.max_token]):
                ids[i] = table.get(word, table["UNK"])
            results["ref_expr_inds"] = ids.numpy() if host_only else ids        # numpy pickles inline across the worker boundary
        results.update(expression=expression, max_token=self.max_token)
        # ---- boxes
        h, w = shape[:2]
        if self.dataset == "GRefCOCO":
            if self.with_bbox:
                results["gt_bbox"] = [_xyxy_clipped(b, h, w) for b in ann["bbox"][self.random_ind]]
            results["target"] = copy.deepcopy(ann["annotations"][self.random_ind])
        elif self.with_bbox:
            results["gt_bbox"] = _xyxy_clipped(ann["bbox"], h, w)
        results.update(with_bbox=self.with_bbox, with_mask=self.with_mask)
        return results

    def __repr__(self):
        return (f"{type(self).__name__}(dataset={self.dataset!r}, max_token={self.max_token}, with_bbox={self.with_bbox}, "
                f"use_token_type={self.use_token_type!r})")
