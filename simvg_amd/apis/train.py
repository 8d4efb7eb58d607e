"""One training epoch (the reference's `train_model`, `simvg/apis/train.py:37-176`, and `set_random_seed`, `:20-34`):
per batch forward_train -> zero_grad -> backward -> global-norm clip (cfg.grad_norm_clip) -> optimizer step -> EMA
update -> statistics, and a log line in the reference's format every cfg.log_interval batches.

Organised around the device rather than around the reference's loop body:
  * the epoch runs on the process's training stream, ordered after everything the caller queued before it (arena
    build, parameter broadcast, checkpoint copies, EMA restore) and handed back in order;
  * data parallel: `simvg_amd.dist.GradReducer` all-reduces the gradient arenas as the hand-sequenced backward leaves
    them (no DistributedDataParallel wrapper); the step's logged scalars cross the ranks as ONE packed all-reduce;
  * nothing reads a device value back except the log line."""
import random
import time

import numpy
import torch

from ..utils import get_root_logger, is_main
from .metrics import BRANCHES, RunningMeans, StepScalars, score_predictions, split_gt, unwrap


def set_random_seed(seed, deterministic=False):
    for seeder in (random.seed, numpy.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if deterministic:   # the flags exist on ROCm builds too (MIOpen); the HIP kernels of this package do not read them
        torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False


def on_train_stream(device, fn):
    """Run fn() on the training stream (simvg_amd.graphs.train_stream): the side stream first waits for the work already
    queued on the caller's stream, and the caller's stream waits for the epoch afterwards.  CPU models run fn() as is."""
    if device.type != "cuda":
        return fn()
    from ..graphs import training_stream
    with training_stream(device):
        return fn()


def _grad_reducer(model):
    from ..dist import GradReducer
    r = getattr(model, "_simvg_grad_reducer", None)
    if r is None:
        r = GradReducer(model)
        object.__setattr__(model, "_simvg_grad_reducer", r)
    return r


class _EpochStats:
    """running means of the losses and of the per-branch metrics, and the reference's log line built from them"""

    def __init__(self, grec):
        self.grec, self.means, self.branches = grec, RunningMeans(), 1

    def absorb(self, step):
        for name, value in step.reduce().items():
            self.means.add(name, value)

    def line(self, epoch, batch, batches, seconds, data_seconds, lr):
        m = self.means.means()
        losses = " ".join("{}:{:.3f}".format(k.split("loss_")[-1], v) for k, v in m.items() if "/" not in k)
        text = (f"train-epoch[{epoch+1}]-[{batch+1}/{batches}] time:{seconds:.2f}, data_time: {data_seconds:.2f}, "
                f"loss:[{losses}], lr:{lr:.6f}, ")
        names = BRANCHES[:self.branches]
        if self.grec:
            return (text + "".join("{}_f1: {:.2f}, ".format(b, m["f1/" + b]) for b in names)
                    + "".join("{}_Nacc: {:.2f}, ".format(b, m["nacc/" + b]) for b in names))
        return text + "".join("{}Acc:{:.2f}, ".format(b, m["det/" + b]) for b in names)

    def loss_means(self):
        return {k: v for k, v in self.means.means().items() if "/" not in k}


def _optimise(cfg, model, optimizer, loss, reducer):
    optimizer.zero_grad()
    if reducer is not None:
        reducer.begin()
    loss.backward()
    if reducer is not None:
        reducer.finish()
    if cfg.grad_norm_clip:
        clip = getattr(optimizer, "clip_grad_norm", None)      # FlatAdam: the same global norm over the flat arenas
        if clip is not None:
            clip(cfg.grad_norm_clip)
        else:
            torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.grad_norm_clip)
    optimizer.step()


def _train_epoch(epoch, cfg, model, model_ema, optimizer, loader):
    from ..datasets import extract_data
    model.train()
    sampler = getattr(loader, "sampler", None)
    if hasattr(sampler, "set_epoch") and (cfg.distributed or getattr(sampler, "reshuffles_single_process", False)):
        sampler.set_epoch(epoch)
    device = next(model.parameters()).device
    reducer = _grad_reducer(model) if cfg.distributed else None
    stats = _EpochStats(grec=cfg["dataset"] == "GRefCOCO")
    nothing = torch.zeros(1, device=device)
    batches = len(loader)
    tick = time.time()
    for batch, raw in enumerate(loader):
        data_seconds = time.time() - tick
        raw = dict(raw)
        if "gt_mask_rle" in raw:
            raise NotImplementedError("mask training is outside this hot path")
        raw.pop("is_crowd", None)
        gt_bbox = None
        if "gt_bbox" in raw:
            raw["gt_bbox"] = split_gt(raw)
            gt_bbox = [g.clone() for g in raw["gt_bbox"]]
        img_metas = unwrap(raw["img_metas"])

        losses, predictions = model(**extract_data(raw, device), rescale=False)
        total = losses.get("loss_total", nothing) + losses.get("loss_det", nothing) + losses.pop("loss_mask", nothing)
        _optimise(cfg, model, optimizer, total, reducer)
        if cfg.ema:
            model_ema.update_params()

        step = StepScalars()
        for name, value in losses.items():
            step.put(name, value)
        with torch.no_grad():
            stats.branches = score_predictions(step, predictions, gt_bbox, img_metas, stats.grec, device)
        stats.absorb(step)

        if is_main() and ((batch + 1) % cfg.log_interval == 0 or batch + 1 == batches):
            get_root_logger().info(stats.line(epoch, batch, batches, time.time() - tick, data_seconds,
                                              optimizer.param_groups[0]["lr"]))
        tick = time.time()
    return stats.loss_means()


def train_model(epoch, cfg, model, model_ema, optimizer, loader):
    device = next(model.parameters()).device
    return on_train_stream(device, lambda: _train_epoch(epoch, cfg, model, model_ema, optimizer, loader))
