"""Checkpoint API with the reference's signatures (`simvg/utils/checkpoint.py:53-148`): `load_checkpoint`,
`load_pretrained_checkpoint`, `save_checkpoint` -- `.pth` files interchange with the reference (keys `state_dict`,
`ema_state_dict`, `optimizer`, `scheduler`, `lr`, `epoch`, `d_acc`, `miou`, `best_d_acc`, `best_miou`, `amp`;
`module.` prefixes stripped on load) and the log lines are the reference's.  The optimizer entry is this build's own
(FlatAdam keeps one state per arena), so an optimizer state written by the reference is skipped with a log line instead
of being mis-loaded."""
import copy
import os
import shutil

import torch

from .distributed import is_main
from .logger import get_root_logger

_PREFIX = "module."
# (checkpoint key, log format) in the order the reference prints them after a load
_REPORT = (("best_d_acc", "best det acc: {:.2f}\n"), ("best_miou", "best mIoU: {:.2f}\n"),
           ("d_acc", "loaded det acc: {:.2f}\n"), ("miou", "loaded mIoU: {:.2f}\n"))


def is_paral_state(state_dict):
    """was this state dict written through a DataParallel / DDP wrapper?"""
    return next(iter(state_dict)).startswith(_PREFIX)


def de_parallel(state_dict):
    return {k[len(_PREFIX):]: v for k, v in state_dict.items()}


def _plain(state_dict):
    return de_parallel(state_dict) if is_paral_state(state_dict) else state_dict


def log_loaded_info(ckpt, load_file):
    """one log record describing the loaded file; returns the best scores it carries (0.0 when absent)"""
    text = [f"loaded checkpoint from {load_file}\n"]
    if ckpt.get("lr") is not None and "epoch" in ckpt:
        text.append(f"epoch: {ckpt['epoch']+1} lr: {ckpt['lr']:.6f}\n")
    text += [fmt.format(ckpt[key]) for key, fmt in _REPORT if key in ckpt]
    get_root_logger().info("".join(text))
    return ckpt.get("best_d_acc", 0.0), ckpt.get("best_miou", 0.0)


def _read(path, model):
    return torch.load(path, map_location=next(model.parameters()).device, weights_only=False)


def load_pretrained_checkpoint(model, model_ema=None, finetune_from=None, amp=False):
    """Fine-tuning start (reference :53-83): non-strict load of `state_dict`, the epoch counter starts over."""
    assert model_ema is None, "We do not use EMA during finetuning."
    ckpt = _read(finetune_from, model)
    missing, unexpected = model.load_state_dict(copy.deepcopy(_plain(ckpt["state_dict"])), strict=False)
    best = (0.0, 0.0)
    if is_main():
        log = get_root_logger()
        log.info("missing keys:{}".format(missing))
        log.info("unexpected keys:{}".format(unexpected))
        best = log_loaded_info(ckpt, finetune_from)
    return (-1,) + tuple(best)


def load_checkpoint(model, model_ema=None, resume_from=None, load_from=None, amp=False, optimizer=None, scheduler=None):
    """-> (start_epoch, best_d_acc, best_miou, has_ema) (reference :86-120).  `resume_from` continues a run (epoch,
    optimizer, scheduler, EMA shadow); `load_from` takes the weights only.  has_ema: the file carried an `ema_state_dict` and it
    was assigned to `model_ema.shadow`; when it is False the caller's shadow still holds whatever it was built from (the reference
    raises a NameError on such a file, :100-101) -- tools/train.py and tools/test.py restart the shadow from the loaded weights."""
    assert resume_from is None or load_from is None
    path = resume_from if resume_from is not None else load_from
    ckpt = _read(path, model)
    weights = _plain(ckpt["state_dict"])
    try:
        model.load_state_dict(weights, strict=True)
    except RuntimeError as e:
        if is_main():
            get_root_logger().info(f"{path}: strict load failed ({str(e).splitlines()[0]} ...); loading the matching keys only")
        model.load_state_dict(weights, strict=False)
    has_ema = model_ema is not None and "ema_state_dict" in ckpt
    if has_ema:
        model_ema.shadow = _plain(ckpt["ema_state_dict"])
    for obj, key in ((optimizer, "optimizer"), (scheduler, "scheduler")):
        if obj is None or ckpt.get(key) is None:
            continue
        try:
            obj.load_state_dict(ckpt[key])
        except (ValueError, KeyError) as e:
            if key != "optimizer":
                raise
            if is_main():      # a per-tensor state written by the reference's torch.optim.Adam
                get_root_logger().info(f"optimizer state in {path} does not match this optimizer ({e}); skipped")
    start_epoch = ckpt["epoch"] if (resume_from is not None and "epoch" in ckpt) else -1
    best = log_loaded_info(ckpt, path) if is_main() else (0.0, 0.0)
    return start_epoch, best[0], best[1], has_ema


def save_checkpoint(work_dir, interval, model, model_ema, optimizer, scheduler, checkpoint):
    """latest.pth on every call, epoch_N.pth every `interval` epochs, det_best.pth / segm_best.pth when the epoch's
    score beats the best so far (reference :123-148); `checkpoint` carries epoch / d_acc / miou / best_* / amp."""
    epoch = checkpoint["epoch"] + 1
    checkpoint.pop("use_fp16", False)
    checkpoint["state_dict"] = model.state_dict()
    checkpoint["optimizer"] = optimizer.state_dict()
    checkpoint["scheduler"] = scheduler.state_dict()
    checkpoint["lr"] = optimizer.param_groups[0]["lr"]
    if model_ema is not None:
        checkpoint["ema_state_dict"] = dict(model_ema.shadow)

    def announce(path):
        if is_main():
            get_root_logger().info(f"saved epoch {epoch} checkpoint at {path}")

    latest = os.path.join(work_dir, "latest.pth")
    torch.save(checkpoint, latest)
    announce(latest)
    if interval > 0 and epoch % interval == 0:
        torch.save(checkpoint, os.path.join(work_dir, f"epoch_{epoch}.pth"))
    for score, best, name in (("d_acc", "best_d_acc", "det_best.pth"), ("miou", "best_miou", "segm_best.pth")):
        if checkpoint[score] > checkpoint[best]:
            target = os.path.join(work_dir, name)
            shutil.copyfile(latest, target)
            announce(target)
