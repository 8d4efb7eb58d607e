"""What `tools/train.py` and `tools/test.py` share: a `Session` owns the process group, the logger, the datasets of a
config, the model on this process's MI355X, and the evaluation pass with its EMA double evaluation.

The command lines, the config semantics, the split names, the log lines and the checkpoint files are the reference's
(`tools/train.py:26-216`, `tools/test.py:19-134`); how the run is put together is this package's:
  * one process per GPU, no DistributedDataParallel wrapper -- replicas are made identical by a broadcast of the flat
    arenas' tensors and kept identical by `simvg_amd.dist.GradReducer` inside `train_model`;
  * the encoder is laid out in its flat arenas BEFORE the optimizer, the EMA or a checkpoint take views of it;
  * the annotation-file datasets of a reference config are read as they are (`simvg_amd/datasets/refsets.py`);
    `data.synthetic=True` swaps every split for RefCOCO-shaped synthetic pairs with the split's own name and geometry."""
import os.path as osp
import time

import torch
import torch.distributed as dist

from .apis import evaluate_model
from .datasets import build_dataloader, build_dataset
from .models import build_model
from .models.utils import ExponentialMovingAverage
from .utils import get_dist_info, get_root_logger, init_dist, is_main

MIXED_VAL_SPLITS = ("val_refcoco_unc", "val_refcocoplus_unc", "val_refcocog_umd", "val_referitgame_berkeley", "val_flickr30k")


def elapsed(since):
    """'<m>m-<s>s' since a time.time() stamp (the format of the reference's epoch / total time lines)"""
    t = int(time.time() - since)
    return "{}m-{}s".format(t // 60, t % 60)


class Session:
    def __init__(self, cfg):
        self.cfg = cfg
        cfg.distributed = cfg.launcher == "pytorch"
        if cfg.distributed:
            init_dist()
        cfg.rank, cfg.world_size = get_dist_info()
        if cfg.use_fp16:
            raise NotImplementedError("use_fp16 (apex O1) is not part of this build: the encoder computes with 16-bit MFMA "
                                      "operands and fp32 master weights already; every reference config sets use_fp16=False")
        self._synthetic_splits()
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.model = self.ema = None

    # ---- logging -------------------------------------------------------------------------------------------------
    def say(self, text):
        if is_main():
            get_root_logger().info(text)

    def open_log(self, log_file):
        if is_main():
            get_root_logger(log_file=log_file).info(self.cfg.pretty_text)

    # ---- data ----------------------------------------------------------------------------------------------------
    def _synthetic_splits(self):
        cfg = self.cfg
        if not cfg.data.get("synthetic", False):
            return
        for name, split in cfg.data.items():
            if isinstance(split, dict) and "pipeline" in split:
                split["synthetic"] = True
                for key, default in (("type", cfg.dataset), ("which_set", name), ("img_size", cfg.get("img_size", 640)),
                                     ("max_token", cfg.get("max_token", 20))):
                    split.setdefault(key, default)

    def validation_splits(self):
        """names of the splits evaluated after a training epoch (tools/train.py:113-121)"""
        data = self.cfg.data
        if self.cfg.dataset == "Mixed":
            return [s for s in MIXED_VAL_SPLITS if getattr(data, s, None)]
        return ["val"]

    def test_splits(self):
        """names of the splits `tools/test.py` reports (:46-60)"""
        data = self.cfg.data
        if self.cfg.dataset == "Mixed":
            return list(MIXED_VAL_SPLITS[:3])
        if hasattr(data, "testA") and hasattr(data, "testB"):
            return ["val", "testA", "testB"]
        return ["val", "test"] if hasattr(data, "test") else ["val"]

    def dataset(self, split):
        return build_dataset(getattr(self.cfg.data, split))

    def loader(self, dataset):
        return build_dataloader(self.cfg, dataset)

    # ---- model ---------------------------------------------------------------------------------------------------
    def build(self, train_set):
        """model on this GPU (vocabulary hand-over as in the reference), arenas laid out, replicas made identical, EMA"""
        cfg = self.cfg
        model = build_model(cfg.model, word_emb=train_set.word_emb, num_token=train_set.num_token).to(self.device)
        model.vis_enc._ensure_engine(self.device)
        if cfg.distributed:
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, 0)
        self.model = model
        self.ema = ExponentialMovingAverage(model, cfg.ema_factor) if cfg.ema else None
        return model

    def fresh_ema(self):
        self.ema = ExponentialMovingAverage(self.model, self.cfg.ema_factor) if self.cfg.ema else None

    # ---- evaluation ------------------------------------------------------------------------------------------------
    def evaluate(self, epoch, loader, announce, announce_ema):
        """live weights, then -- with cfg.ema -- the shadow weights (apply_shadow -> evaluate -> restore);
        -> {'': (d_acc, miou)[, '_ema': (d_acc, miou)]}"""
        self.say(announce)
        out = {"": evaluate_model(epoch, self.cfg, self.model, loader)}
        if self.cfg.ema:
            self.say(announce_ema)
            self.ema.apply_shadow()
            try:
                out["_ema"] = evaluate_model(epoch, self.cfg, self.model, loader)
            finally:
                self.ema.restore()
        return out

    def close(self):
        if self.cfg.distributed:
            dist.destroy_process_group()
