"""Checkpoint API with the reference's signatures (`simvg/utils/checkpoint.py:53-148`): `load_checkpoint`,
`load_pretrained_checkpoint`, `save_checkpoint` -- `.pth` files interchange with the reference (keys `state_dict`,
`ema_state_dict`, `optimizer`, `scheduler`, `lr`, `epoch`, `d_acc`, `miou`, `best_d_acc`, `best_miou`, `amp`;
`module.` prefixes stripped on load).  The optimizer entry is this build's own (FlatAdam keeps one state per arena),
so an optimizer state written by the reference is skipped with a log line instead of being mis-loaded."""
import copy
import os.path as osp
import shutil

import torch

from .distributed import is_main
from .logger import get_root_logger


def is_paral_state(state_dict):
    return list(state_dict.keys())[0].startswith("module.")


def de_parallel(state_dict):
    return {key[7:]: value for key, value in state_dict.items()}


def log_loaded_info(ckpt, load_file):
    logger = get_root_logger()
    log_str = f"loaded checkpoint from {load_file}\n"
    best_d_acc, best_miou = 0.0, 0.0
    if "lr" in ckpt and ckpt["lr"] is not None and "epoch" in ckpt:
        log_str += f"epoch: {ckpt['epoch']+1} lr: {ckpt['lr']:.6f}\n"
    if "best_d_acc" in ckpt:
        log_str += f"best det acc: {ckpt['best_d_acc']:.2f}\n"
        best_d_acc = ckpt["best_d_acc"]
    if "best_miou" in ckpt:
        log_str += f"best mIoU: {ckpt['best_miou']:.2f}\n"
        best_miou = ckpt["best_miou"]
    if "d_acc" in ckpt:
        log_str += f"loaded det acc: {ckpt['d_acc']:.2f}\n"
    if "miou" in ckpt:
        log_str += f"loaded mIoU: {ckpt['miou']:.2f}\n"
    logger.info(log_str)
    return best_d_acc, best_miou


def _device_of(model):
    return next(model.parameters()).device


def load_pretrained_checkpoint(model, model_ema=None, finetune_from=None, amp=False):
    """Fine-tuning start: non-strict load of `state_dict`, epoch counter reset (reference :53-83)."""
    assert model_ema is None, "We do not use EMA during finetuning."
    start_epoch, best_d_acc, best_miou = -1, 0.0, 0.0
    ckpt = torch.load(finetune_from, map_location=_device_of(model), weights_only=False)
    state = ckpt["state_dict"]
    if is_paral_state(state):
        state = de_parallel(state)
    missing_keys, unexpected_keys = model.load_state_dict(copy.deepcopy(state), strict=False)
    if is_main():
        logger = get_root_logger()
        logger.info("missing keys:{}".format(missing_keys))
        logger.info("unexpected keys:{}".format(unexpected_keys))
        best_d_acc, best_miou = log_loaded_info(ckpt, finetune_from)
    return start_epoch, best_d_acc, best_miou


def load_checkpoint(model, model_ema=None, resume_from=None, load_from=None, amp=False, optimizer=None, scheduler=None):
    """-> (start_epoch, best_d_acc, best_miou, strict_ok) (reference :86-120)."""
    start_epoch, best_d_acc, best_miou = -1, 0.0, 0.0
    flag = True
    assert not (resume_from is not None and load_from is not None)
    load_file = resume_from or load_from
    ckpt = torch.load(load_file, map_location=_device_of(model), weights_only=False)
    state = ckpt["state_dict"]
    ema_state = None
    if "ema_state_dict" in ckpt:
        ema_state = ckpt["ema_state_dict"]
        if is_paral_state(ema_state):
            ema_state = de_parallel(ema_state)
    if is_paral_state(state):
        state = de_parallel(state)
    try:
        model.load_state_dict(state, strict=True)
    except RuntimeError:
        model.load_state_dict(state, strict=False)
        flag = False
    if model_ema is not None and ema_state is not None:
        model_ema.shadow = ema_state
    if optimizer is not None and ckpt.get("optimizer") is not None:
        try:
            optimizer.load_state_dict(ckpt["optimizer"])
        except (ValueError, KeyError) as e:   # a per-tensor state written by the reference's torch.optim.Adam
            if is_main():
                get_root_logger().info(f"optimizer state in {load_file} does not match this optimizer ({e}); skipped")
    if scheduler is not None and ckpt.get("scheduler") is not None:
        scheduler.load_state_dict(ckpt["scheduler"])
    if "epoch" in ckpt and load_from is None and resume_from is not None:
        start_epoch = ckpt["epoch"]
    if is_main():
        best_d_acc, best_miou = log_loaded_info(ckpt, load_file)
    return start_epoch, best_d_acc, best_miou, flag


def save_checkpoint(work_dir, interval, model, model_ema, optimizer, scheduler, checkpoint):
    """latest.pth every call, epoch_N.pth every `interval`, det_best / segm_best copies on improvement (:123-148)."""
    epoch = checkpoint["epoch"] + 1
    logger = get_root_logger()
    checkpoint.pop("use_fp16", False)
    checkpoint.update({
        "state_dict": model.state_dict(),
        "optimizer": optimizer.state_dict(),
        "scheduler": scheduler.state_dict(),
        "lr": optimizer.param_groups[0]["lr"],
    })
    if model_ema is not None:
        checkpoint.update({"ema_state_dict": dict(model_ema.shadow)})
    latest_path = osp.join(work_dir, "latest.pth")
    det_best_path = osp.join(work_dir, "det_best.pth")
    segm_best_path = osp.join(work_dir, "segm_best.pth")
    torch.save(checkpoint, latest_path)
    if is_main():
        logger.info(f"saved epoch {epoch} checkpoint at {latest_path}")
    if interval > 0 and epoch % interval == 0:
        torch.save(checkpoint, osp.join(work_dir, f"epoch_{epoch}.pth"))
    if checkpoint["d_acc"] > checkpoint["best_d_acc"]:
        shutil.copyfile(latest_path, det_best_path)
        if is_main():
            logger.info(f"saved epoch {epoch} checkpoint at {det_best_path}")
    if checkpoint["miou"] > checkpoint["best_miou"]:
        shutil.copyfile(latest_path, segm_best_path)
        if is_main():
            logger.info(f"saved epoch {epoch} checkpoint at {segm_best_path}")
