"""Data side of the drop-in tools: registries + loader construction with the reference's call shape
(`simvg/datasets/builder.py:16-58`: `build_dataset(cfg.data.train)`, `build_dataloader(cfg, dataset)`,
`extract_data(inputs)`).

Two sources:
  * the reference's annotation-file datasets under their own names (`refsets.py`: RefCOCOUNC ... GRefCOCO, Mixed), whose
    pipeline starts with `LoadImageAnnotationsFromFile` (`loading.py`: file naming, decode, expression choice, XLM-R
    sentencepiece ids, boxes) and continues with the DEVICE-side transforms of `pipelines.py` (LargeScaleJitter / Resize
    / Normalize / Pad / DefaultFormatBundle / CollectData as HIP kernels on a uint8 frame in HBM);
  * `SyntheticRefDataset`, RefCOCO-shaped random pairs (this image holds no dataset and has no network): selected with
    `type="SyntheticRefDataset"` or globally with `--cfg-options data.synthetic=True`, which keeps every other key of a
    reference config (pipelines, annsfile, ...) untouched and ignores them."""
import functools

import torch
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

from ..models.builder import Registry

DATASETS = Registry("DATASETS")
PIPELINES = Registry("PIPELINES")

_REFERENCE_DATASETS = ("RefCOCOUNC", "RefCOCOGoogle", "RefCOCOgUMD", "RefCOCOgGoogle", "RefCOCOPlusUNC",
                       "ReferItGameBerkeley", "Flickr30k", "Mixed", "GRefCOCO", "MixedSeg", "RefClef")


@DATASETS.register_module()
class SyntheticRefDataset(Dataset):
    """`length` (image, expression, box) triples of RefCOCO shape: 640x640 normalised image noise, XLM-R style ids
    (<s> tokens </s> pad...) with their padding mask, one xyxy box in pixels (GRefCOCO: `max_targets` boxes, some
    images with no target).  Sample i is a pure function of (seed, i): every rank / epoch sees reproducible data."""

    def __init__(self, which_set="train", length=256, img_size=640, max_token=20, vocab_size=64010, seed=0,
                 dataset="RefCOCOUNC", max_targets=1, **ignored):
        self.which_set, self.length, self.img_size = which_set, int(length), int(img_size)
        self.max_token, self.vocab_size, self.seed = int(max_token), int(vocab_size), int(seed)
        self.grec = dataset == "GRefCOCO"
        self.max_targets = max(int(max_targets), 1) if self.grec else 1
        self.word_emb, self.num_token = None, -1          # what tools/train.py hands to build_model
        self.flag = torch.zeros(self.length, dtype=torch.uint8).numpy()   # GroupSampler aspect-ratio groups: one group

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        S, T = self.img_size, self.max_token
        img = torch.randn(3, S, S, generator=g)
        ids = torch.ones(T, dtype=torch.int64)
        pad = torch.ones(T, dtype=torch.int64)
        m = int(torch.randint(2, T - 1, (1,), generator=g))
        ids[0] = 0
        ids[1:1 + m] = torch.randint(4, self.vocab_size, (m,), generator=g)
        ids[1 + m] = 2
        pad[:m + 2] = 0
        k = 1 if not self.grec else int(torch.randint(0, self.max_targets + 1, (1,), generator=g))
        n = max(k, 1)
        xy = torch.rand(n, 2, generator=g) * (S * 0.6)
        wh = S * 0.05 + torch.rand(n, 2, generator=g) * (S * 0.3)
        boxes = torch.cat([xy, xy + wh], 1)
        meta = dict(img_shape=(S, S, 3), pad_shape=(S, S, 3), ori_shape=(S, S, 3), scale_factor=[1.0] * 4,
                    filename=f"synthetic_{self.which_set}_{i}.jpg", expression="synthetic")
        if self.grec:
            if k == 0:
                boxes = torch.zeros(1, 4)
            meta["target"] = [dict(category_id=-1 if k == 0 else 1) for _ in range(n)]
            gt = boxes
        else:
            gt = boxes[0]
        return dict(img=img, ref_expr_inds=ids, text_attention_mask=pad, gt_bbox=gt, img_metas=meta)


from . import pipelines as _pipelines   # noqa: E402,F401  (registers the device-side transforms in PIPELINES)
from . import loading as _loading       # noqa: E402,F401  (LoadImageAnnotationsFromFile)
from . import refsets as _refsets       # noqa: E402,F401  (the file-backed datasets under the reference's names)
from .refsets import AspectGroupSampler  # noqa: E402


def _stack_images(imgs):
    """[C, h, w] tensors -> [B, C, H, W], zero-padded at the bottom / right to the largest frame of the batch (the
    reference's collate pads the last two dims, `datasets/utils.py:76-100`)"""
    H, W = max(int(i.shape[-2]) for i in imgs), max(int(i.shape[-1]) for i in imgs)
    if all(tuple(i.shape[-2:]) == (H, W) for i in imgs):
        return torch.stack(imgs)
    out = imgs[0].new_zeros((len(imgs), imgs[0].shape[0], H, W))
    for k, i in enumerate(imgs):
        out[k, :, :i.shape[-2], :i.shape[-1]] = i
    return out


def _collate(batch, stacked_img=None):
    """stacked_img: the frames already as one [B, C, H, W] tensor (the batched device stage writes them in place)"""
    img = stacked_img if stacked_img is not None else _stack_images([b["img"] for b in batch])
    out = dict(img=img, img_metas=[b["img_metas"] for b in batch])
    for key in ("ref_expr_inds", "text_attention_mask"):          # the mask exists for tokenizer ids only
        if key in batch[0]:
            out[key] = torch.stack([torch.as_tensor(b[key]) for b in batch])
    if "gt_bbox" in batch[0]:
        gts = [b["gt_bbox"] for b in batch]
        out["gt_bbox"] = torch.stack(gts) if all(g.dim() == 1 for g in gts) else gts
    return out


def build_dataset(cfg, default_args=None):
    cfg = dict(cfg)
    typ = cfg.get("type")
    synthetic = cfg.pop("synthetic", False)
    if typ in _REFERENCE_DATASETS and synthetic:
        keep = {k: cfg[k] for k in ("which_set", "length", "img_size", "max_token", "seed", "max_targets") if k in cfg}
        cfg = dict(type="SyntheticRefDataset", dataset=typ, **keep)
    elif typ in ("MixedSeg", "RefClef"):
        raise NotImplementedError(f"dataset type {typ!r} (segmentation targets) is not built; see datasets/refsets.py for the "
                                  "file-backed detection datasets or pass --cfg-options data.synthetic=True")
    elif typ in _REFERENCE_DATASETS:
        for k in ("length", "img_size", "max_token", "seed", "max_targets"):      # keys of the synthetic source only
            cfg.pop(k, None)
    return DATASETS.build(cfg, default_args=default_args)


def build_dataloader(cfg, dataset):
    sampler, shuffle = None, False
    train = dataset.which_set == "train"
    grouped = train and isinstance(dataset, _refsets.RefFileDataset)      # aspect-ratio batches for real frames
    if grouped:
        sampler = AspectGroupSampler(dataset.flag, cfg.data.samples_per_gpu, cfg.world_size if cfg.distributed else 1,
                                     cfg.rank if cfg.distributed else 0, seed=cfg.seed or 0)
    elif cfg.distributed:
        sampler = DistributedSampler(dataset, cfg.world_size, cfg.rank, shuffle=train, seed=cfg.seed or 0)
    elif train:
        shuffle = True
    g = torch.Generator()
    g.manual_seed(cfg.seed or 0)
    if isinstance(dataset, _refsets.RefFileDataset) and dataset.host_steps() > 0:
        # file read + JPEG decode + tokenisation in `workers_per_gpu` worker processes (seeded like the reference's:
        # num_workers * rank + worker_id + seed), every pixel transform on the GPU in this process
        workers = int(cfg.data.get("workers_per_gpu", 0) or 0)
        rank, seed = (cfg.rank if cfg.distributed else 0), cfg.seed
        init = functools.partial(_seed_worker, num_workers=workers, rank=rank, seed=seed)
        host = DataLoader(_refsets.HostStageView(dataset), batch_size=cfg.data.samples_per_gpu, sampler=sampler, shuffle=shuffle,
                          generator=g, num_workers=workers, pin_memory=bool(cfg.data.get("pin_memory", False)),
                          collate_fn=_refsets.pack_host_batch, worker_init_fn=init, drop_last=False, persistent_workers=workers > 0)
        return _refsets.TwoStageLoader(dataset, host, _collate)
    return DataLoader(dataset, batch_size=cfg.data.samples_per_gpu, sampler=sampler, shuffle=shuffle, generator=g,
                      num_workers=0, pin_memory=False, collate_fn=_collate, drop_last=False)


def _seed_worker(worker_id, num_workers, rank, seed):
    import random
    import numpy
    torch.set_num_threads(1)           # a worker decodes and copies one frame at a time: no intra-op thread pool per worker
    if seed is not None:
        s = num_workers * rank + worker_id + seed
        numpy.random.seed(s)
        random.seed(s)


def extract_data(inputs, device=None):
    """dict of batch entries -> what the model's forward takes, tensors on `device` (the reference's version unwraps
    mmcv DataContainers and scatters to the current GPU, `datasets/utils.py:38-52`)."""
    assert isinstance(inputs, dict)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    out = {}
    for key, value in inputs.items():
        if hasattr(value, "data") and not isinstance(value, torch.Tensor):      # DataContainer-like
            value = value.data[0] if getattr(value, "cpu_only", False) or not isinstance(value.data, torch.Tensor) else value.data
        if isinstance(value, torch.Tensor):
            value = value.to(device, non_blocking=True)
        elif isinstance(value, (list, tuple)) and value and all(isinstance(v, torch.Tensor) for v in value):
            value = [v.to(device, non_blocking=True) for v in value]
        out[key] = value
    return out
