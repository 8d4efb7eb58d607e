"""Annotation-file datasets with the reference's registry names and constructor arguments (`simvg/datasets/base.py:13-175`:
RefCOCOUNC, RefCOCOGoogle, RefCOCOgUMD, RefCOCOgGoogle, RefCOCOPlusUNC, ReferItGameBerkeley, Flickr30k, Mixed, GRefCOCO).

One annotation json holds every split: {"train": [record...], "val": [...], ...}; a record carries image_id, width, height,
expressions, bbox (xywh; GRefCOCO: one list of boxes and one list of `annotations` per expression) and, for `Mixed`,
data_source.  A dataset item = the split's record pushed through the config's pipeline (first step
`LoadImageAnnotationsFromFile`, then the device transforms).  The word vocabulary (`token2idx`, handed to `build_model` as
num_token / word_emb by tools/train.py) is built by scanning all expressions in file order, or read from the
token_to_ix.pkl / ix_to_token.pkl / word_emb.npz cache the reference leaves beside the json (`datasets/utils.py:136-190`)."""
import json
import os.path as osp
import pickle

import numpy
import torch
from torch.utils.data import Dataset, Sampler

from . import DATASETS
from .loading import clean_string

SPLITS = ("train", "val", "testA", "testB", "test", "val_refcoco_unc", "val_refcocoplus_unc", "val_refcocog_umd",
          "val_flickr30k", "val_referitgame_berkeley")
IMAGE_SOURCES = ("coco", "visual-genome", "flickr", "saiaprtc12")


def build_vocabulary(annsfile, anns_all, word_emb_cfg=None, write_cache=False):
    """-> (token2idx, idx2token, word_emb).  Ids follow first occurrence over the splits in json order after PAD 0 / UNK 1 /
    CLS 2.  GloVe vectors need spacy's en_vectors_web_lg; the BEiT-3 models take no word embedding (lan_enc=None), so a
    missing package yields an empty `word_emb` instead of the reference's ImportError."""
    base = osp.dirname(annsfile)
    paths = [osp.join(base, n) for n in ("token_to_ix.pkl", "ix_to_token.pkl", "word_emb.npz")]
    if all(osp.exists(p) for p in paths):
        with open(paths[0], "rb") as f:
            token2idx = pickle.load(f)
        with open(paths[1], "rb") as f:
            idx2token = pickle.load(f)
        return token2idx, idx2token, numpy.load(paths[2], allow_pickle=True)["word_emb"]
    vectors = None
    if word_emb_cfg is not None and dict(word_emb_cfg).get("type") == "GloVe":
        try:
            import en_vectors_web_lg
            vectors = en_vectors_web_lg.load()
        except ImportError:
            vectors = None
    token2idx = {"PAD": 0, "UNK": 1, "CLS": 2}
    for records in anns_all.values():
        for record in records:
            for expression in record["expressions"]:
                for word in clean_string(expression).split():
                    token2idx.setdefault(word, len(token2idx))
    idx2token = {i: t for t, i in token2idx.items()}
    word_emb = numpy.array([vectors(t).vector for t in token2idx]) if vectors is not None else numpy.array([])
    if write_cache:
        with open(paths[0], "wb") as f:
            pickle.dump(token2idx, f, protocol=pickle.HIGHEST_PROTOCOL)
        with open(paths[1], "wb") as f:
            pickle.dump(idx2token, f, protocol=pickle.HIGHEST_PROTOCOL)
        numpy.savez_compressed(paths[2], word_emb=word_emb)
    return token2idx, idx2token, word_emb


class RefFileDataset(Dataset):
    """imgsfile: image directory (one source) or {source: directory} (several, `Mixed`)"""

    def __init__(self, imgsfile, annsfile, pipeline, which_set="train", img_source=("coco",), word_emb_cfg=None,
                 write_vocab_cache=False):
        from .pipelines import Compose
        if which_set not in SPLITS:
            raise ValueError(f"which_set must be one of {SPLITS}")
        img_source = list(img_source)
        if not img_source:
            raise TypeError("img_source should be a list of str")
        if len(img_source) == 1:
            if img_source[0] not in IMAGE_SOURCES:
                raise ValueError(f"unknown image source {img_source[0]!r}")
        elif not (isinstance(imgsfile, dict) and len(imgsfile) == len(img_source)):
            raise ValueError("several image sources need imgsfile = {source: directory}")
        self.which_set, self.imgsfile = which_set, imgsfile
        with open(annsfile, "r") as f:
            self.anns_all = json.load(f)
        self.token2idx, self.idx2token, self.word_emb = build_vocabulary(annsfile, self.anns_all, word_emb_cfg, write_vocab_cache)
        train = self.anns_all.get("train", [])
        if train and train[0].get("data_source") is not None:
            self.anns_all["train"] = [r for r in train if r["data_source"] in img_source]
        self.records = self.anns_all[which_set]
        if which_set == "train":       # aspect-ratio groups of the group sampler: 1 = landscape
            self.flag = numpy.array([1 if r["width"] / r["height"] > 1 else 0 for r in self.records], dtype=numpy.uint8)
        self.pipeline = Compose(pipeline)
        head = self.pipeline.transforms[0]
        if getattr(head, "use_token_type", "default") == "beit3":
            self.num_token = -1        # the encoder owns its 64 010-row table: build_model ignores num_token
        else:
            self.num_token = len(self.token2idx)

    def __len__(self):
        return len(self.records)

    def _record(self, index):
        return dict(ann=self.records[index], which_set=self.which_set, token2idx=self.token2idx, imgsfile=self.imgsfile)

    def __getitem__(self, index):
        return self.pipeline(self._record(index))

    # ---- two-stage form used by build_dataloader: the host stage runs in DataLoader worker processes -- file read, JPEG
    # decode, tokenisation and, when every later transform can record its pixel work on a DeferredFrame, the WHOLE pipeline's
    # host logic (random draws, crop search, box arithmetic) --, the device stage in the training process on the GPU
    def host_steps(self):
        ts = self.pipeline.transforms
        n = 0
        while n < len(ts) and getattr(ts[n], "host_side", False):
            n += 1
        if n and all(getattr(t, "deferrable", False) for t in ts[n:]):
            return len(ts)
        return n

    def host_item(self, index):
        from .pipelines import DeferredFrame
        results = self._record(index)
        ts = self.pipeline.transforms
        n = self.host_steps()
        for k, t in enumerate(ts[:n]):
            if getattr(t, "host_side", False):
                results = t(results, host_only=True)
                if n == len(ts) and torch.is_tensor(results.get("img")):
                    results["img"] = DeferredFrame(results["img"])
            else:
                results = t(results)
        return results

    def device_item(self, results, device, packed=None):
        img = results.get("img")
        if hasattr(img, "materialize"):
            from .pipelines import tensorize
            results["img"] = img.materialize(device, packed if img.offset is not None else None)
            tensorize(results)
        elif torch.is_tensor(img):
            results["img"] = img.to(device, non_blocking=True)
        for t in self.pipeline.transforms[self.host_steps():]:
            results = t(results)
        return results


class HostStageView(Dataset):
    """what the DataLoader's workers see of a RefFileDataset: records -> decoded uint8 frame (CPU) + ids + boxes"""

    def __init__(self, dataset):
        self.dataset = dataset

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index):
        out = self.dataset.host_item(index)
        out.pop("token2idx", None)           # the vocabulary table does not need to travel back from the worker
        return out


def pack_host_batch(items):
    """collate of the HOST stage (runs in the worker).  The batch's decoded frames become ONE flat uint8 tensor -- one
    shared-memory hand-over and one host-to-device transfer per batch instead of one per frame --, their recorded pixel
    operations are compiled into launch descriptors (pipelines.compile_batch_program), ids / masks / single boxes are stacked:
    what reaches the training process is a handful of arrays, and its share of the work is a copy, three to five launches
    and no per-frame Python."""
    from .pipelines import compile_batch_program
    frames = [it["img"] for it in items if hasattr(it.get("img"), "materialize") and it["img"].pixels is not None]
    if not frames or len(frames) != len(items):
        return dict(items=items, frames=None, program=None)
    blob = torch.empty(sum(f.pixels.numel() for f in frames), dtype=torch.uint8)
    at = 0
    for f in frames:
        n = f.pixels.numel()
        blob[at:at + n] = f.pixels.reshape(-1)
        f.offset, f.pixels = at, None
        at += n
    program = compile_batch_program(frames)
    if program is None:
        return dict(items=items, frames=blob, program=None)
    fields = {}
    for key in items[0]:
        if key in ("img", "img_metas"):
            continue
        vals = [numpy.asarray(it[key]) for it in items]
        fields[key] = numpy.stack(vals) if (key != "gt_bbox" or all(v.ndim == 1 for v in vals)) else vals
    return dict(items=None, frames=blob, program=program, fields=fields, metas=[it["img_metas"] for it in items])


class TwoStageLoader:
    """DataLoader over the host stage (optionally in worker processes) + the device stage and the collate in the consumer.
    Presents what `train_model` / `evaluate_model` use of a DataLoader: iteration, len(), .dataset, .sampler.
    With worker processes and a GPU the device stage of batch k+1 (one host-to-device copy, 2-3 kernel launches per frame,
    the stack) runs in a background thread on its own HIP stream while the training step of batch k is being enqueued;
    the consumer's stream waits on the batch's event."""

    def __init__(self, dataset, host_loader, collate, device=None, background=None):
        self.dataset, self.host_loader, self.collate, self.device = dataset, host_loader, collate, device
        self.sampler = host_loader.sampler
        self.batch_size = host_loader.batch_size
        self.background = background

    def __len__(self):
        return len(self.host_loader)

    @staticmethod
    def _to_device(batch, device):
        """ids / masks / boxes of the collated batch -> HBM here, i.e. on the loader's stream: a pageable host-to-device copy
        issued later on the TRAINING stream is synchronous in stream order -- the host would wait for the previous step to
        drain before it could enqueue the next one (measured: 6 ms of GPU idle per step)."""
        if device.type != "cuda":
            return batch
        for k, v in batch.items():
            if torch.is_tensor(v):
                batch[k] = v.to(device, non_blocking=True)
            elif isinstance(v, (list, tuple)) and v and all(torch.is_tensor(t) for t in v):
                batch[k] = [t.to(device, non_blocking=True) for t in v]
        return batch

    def _batches(self, device):
        for batch in self._host_batches(device):
            yield self._to_device(batch, device)

    def _host_batches(self, device):
        from .pipelines import materialize_batch, tensorize
        from .pipelines import run_batch_program
        for host_batch in self.host_loader:
            packed, items = host_batch["frames"], host_batch["items"]
            if host_batch.get("program") is not None and device.type == "cuda":
                # the worker compiled the batch: one copy, a few batched launches, ready-made arrays
                batch = dict(img=run_batch_program(host_batch["program"], packed.to(device, non_blocking=True), device),
                             img_metas=host_batch["metas"])
                for key, v in host_batch["fields"].items():
                    batch[key] = [torch.from_numpy(x) for x in v] if isinstance(v, list) else torch.from_numpy(v)
                yield batch
                continue
            if host_batch.get("program") is not None:        # no GPU (CPU tests): back to per-frame form
                raise RuntimeError("a compiled batch program needs the GPU; build the loader with device transforms on a GPU box")
            if packed is not None and device.type == "cuda" and all(getattr(it.get("img"), "offset", None) is not None for it in items):
                # the whole batch's pixel work in a handful of launches (pipelines.materialize_batch)
                packed = packed.to(device, non_blocking=True)
                imgs, stacked = materialize_batch([it["img"] for it in items], packed, device)
                for it, img in zip(items, imgs):
                    it["img"] = img
                    tensorize(it)
                yield self.collate(items, stacked) if stacked is not None else self.collate(items)
                continue
            if packed is not None:
                packed = packed.to(device, non_blocking=True)
            yield self.collate([self.dataset.device_item(r, device, packed) for r in items])

    def __iter__(self):
        device = self.device
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        background = self.background
        if background is None:
            background = device.type == "cuda" and self.host_loader.num_workers > 0
        if not background:
            yield from self._batches(device)
            return
        import queue
        import threading
        ready, stop = queue.Queue(maxsize=2), threading.Event()
        side = torch.cuda.Stream(device)

        def hand_over(item):                      # False: the consumer is gone
            while not stop.is_set():
                try:
                    ready.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def produce():
            try:
                with torch.cuda.stream(side):
                    for batch in self._batches(device):
                        done = torch.cuda.Event()
                        done.record(side)
                        if not hand_over((batch, done)):
                            return
                hand_over(None)
            except BaseException as err:          # surfaces in the consumer
                hand_over(err)

        worker = threading.Thread(target=produce, name="simvg-device-stage", daemon=True)
        worker.start()
        try:
            import time
            self.wait_s, self.first_wait_s = 0.0, None    # time the consumer spent waiting for finished batches (diagnostic)
            while True:
                t0 = time.perf_counter()
                got = ready.get()
                if self.first_wait_s is None:      # the epoch's pipeline fill (workers start decoding when iteration begins)
                    self.first_wait_s = time.perf_counter() - t0
                else:
                    self.wait_s += time.perf_counter() - t0
                if got is None:
                    break
                if isinstance(got, BaseException):
                    raise got
                batch, done = got
                current = torch.cuda.current_stream(device)
                current.wait_event(done)
                for v in batch.values():           # allocated on the side stream, consumed on this one
                    for t in (v if isinstance(v, (list, tuple)) else [v]):
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(current)
                yield batch
        finally:
            stop.set()


def _register(name):
    cls = type(name, (RefFileDataset,), {"__doc__": f"`{name}` of the reference's DATASETS registry (file-backed)"})
    DATASETS.register_module()(cls)
    return cls


FILE_DATASETS = {n: _register(n) for n in ("GRefCOCO", "RefCOCOUNC", "RefCOCOGoogle", "RefCOCOgUMD", "RefCOCOgGoogle",
                                           "RefCOCOPlusUNC", "ReferItGameBerkeley", "Flickr30k", "Mixed")}


class AspectGroupSampler(Sampler):
    """Batches of `samples_per_gpu` indices that share an aspect-ratio group (the role of mmdet's GroupSampler /
    DistributedGroupSampler in `datasets/builder.py:31-40`): each group is shuffled and padded (by re-drawing its own
    members) to a multiple of the batch -- of batch x world for the distributed form --, the batches of all groups are
    shuffled, and rank r takes every world-th batch.  Reshuffles per epoch through `set_epoch`."""

    reshuffles_single_process = True       # train_model calls set_epoch for it also without torch.distributed

    def __init__(self, flags, samples_per_gpu, world_size=1, rank=0, seed=0):
        self.flags = numpy.asarray(flags)
        self.batch, self.world, self.rank, self.seed, self.epoch = int(samples_per_gpu), int(world_size), int(rank), int(seed or 0), 0
        chunk = self.batch * self.world
        self._len = sum(-(-int(n) // chunk) * chunk for n in numpy.bincount(self.flags) if n) // self.world

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        return self._len

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        chunk = self.batch * self.world
        batches = []
        for flag in numpy.unique(self.flags):
            members = torch.from_numpy(numpy.where(self.flags == flag)[0])
            order = members[torch.randperm(len(members), generator=g)]
            short = -len(order) % chunk
            if short:
                order = torch.cat([order, order[torch.randint(len(order), (short,), generator=g)]])
            batches += list(order.view(-1, self.batch))
        pick = torch.randperm(len(batches) // self.world, generator=g)
        mine = [batches[int(i) * self.world + self.rank] for i in pick]
        return iter(torch.cat(mine).tolist()) if mine else iter(())
