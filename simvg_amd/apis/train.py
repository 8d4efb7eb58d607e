"""Training epoch -- mirror of the reference's `simvg/apis/train.py` (`set_random_seed` :20-34, `train_model`
:37-176): forward_train -> zero_grad -> backward -> clip_grad_norm(cfg.grad_norm_clip) -> optimizer.step ->
EMA update -> running statistics -> log line every cfg.log_interval batches, same line format.

Differences that are deliberate (MI355X-first, same results):
  * data parallel: instead of wrapping the model in (MM)DistributedDataParallel, `simvg_amd.dist.GradReducer` all-reduces
    the flat gradient arena slice of encoder layer i over RCCL the moment the hand-sequenced backward leaves it;
  * the clip runs over the flat arenas when the optimizer offers `clip_grad_norm` (FlatAdam) -- the same global norm;
  * statistics stay on the device between log lines (no per-batch `.item()` stall)."""
import random
import time

import numpy
import torch

from ..utils import get_root_logger, reduce_mean, is_main
from .test import accuracy, grec_evaluate_f1_nacc, RunningMeans, MAP_DICT, _unwrap, _split_gt


def set_random_seed(seed, deterministic=False):
    random.seed(seed)
    numpy.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if deterministic:   # the flags exist on ROCm builds too (MIOpen); harmless for the HIP kernels of this package
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False


def _reducer_of(model):
    from ..dist import GradReducer
    r = getattr(model, "_simvg_grad_reducer", None)
    if r is None:
        r = GradReducer(model)
        object.__setattr__(model, "_simvg_grad_reducer", r)
    return r


def train_model(epoch, cfg, model, model_ema, optimizer, loader):
    device = next(model.parameters()).device
    if device.type == "cuda":      # non-default stream: precondition of the head's hipGraph replay (simvg_amd/graphs.py)
        from ..graphs import train_stream
        with torch.cuda.stream(train_stream(device)):
            out = _train_epoch(epoch, cfg, model, model_ema, optimizer, loader)
        torch.cuda.current_stream(device).wait_stream(train_stream(device))
        return out
    return _train_epoch(epoch, cfg, model, model_ema, optimizer, loader)


def _train_epoch(epoch, cfg, model, model_ema, optimizer, loader):
    from ..datasets import extract_data
    model.train()
    if cfg.distributed and hasattr(getattr(loader, "sampler", None), "set_epoch"):
        loader.sampler.set_epoch(epoch)
    device = next(model.parameters()).device
    batches = len(loader)
    end = time.time()
    grec = cfg["dataset"] == "GRefCOCO"
    loss_stat, det, f1, nacc = RunningMeans(), RunningMeans(), RunningMeans(), RunningMeans()
    reducer = _reducer_of(model) if cfg.distributed else None
    zero = torch.zeros(1, device=device)
    for batch, inputs in enumerate(loader):
        data_time = time.time() - end
        inputs = dict(inputs)
        gt_bbox = None
        if "gt_bbox" in inputs:
            inputs["gt_bbox"] = _split_gt(inputs)
            gt_bbox = [g.clone() for g in inputs["gt_bbox"]]
        img_metas = _unwrap(inputs["img_metas"])
        if "gt_mask_rle" in inputs:
            raise NotImplementedError("mask training is outside this hot path")
        inputs.pop("is_crowd", None)
        inputs = extract_data(inputs, device)

        losses, predictions = model(**inputs, rescale=False)

        loss_det = losses.get("loss_total", zero) + losses.get("loss_det", zero)
        loss = loss_det + losses.pop("loss_mask", zero)
        optimizer.zero_grad()
        if reducer is not None:
            reducer.begin()
        loss.backward()
        if reducer is not None:
            reducer.finish()
        if cfg.grad_norm_clip:
            if hasattr(optimizer, "clip_grad_norm"):
                optimizer.clip_grad_norm(cfg.grad_norm_clip)
            else:
                torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.grad_norm_clip)
        optimizer.step()
        if cfg.ema:
            model_ema.update_params()

        predictions_list = predictions if isinstance(predictions, list) else [predictions]
        for loss_name, loss_value in losses.items():
            if cfg.distributed:
                loss_value = reduce_mean(loss_value.detach())
            loss_stat.add(loss_name, loss_value)
        for ind, predictions in enumerate(predictions_list):
            predict_type = MAP_DICT[ind]
            pred_bboxes = predictions.pop("pred_bboxes")
            pred_masks = predictions.pop("pred_masks")
            with torch.no_grad():
                if not grec:
                    batch_det_acc, _, _ = accuracy(pred_bboxes, [g.to(device) for g in gt_bbox], pred_masks, None, device=device)
                    if cfg.distributed:
                        batch_det_acc = reduce_mean(batch_det_acc)
                    det.add(predict_type, batch_det_acc)
                else:
                    targets = [meta["target"] for meta in img_metas]
                    batch_f1_score, batch_n_acc = grec_evaluate_f1_nacc(pred_bboxes, gt_bbox, targets, device=device)
                    if cfg.distributed:
                        batch_f1_score, batch_n_acc = reduce_mean(batch_f1_score), reduce_mean(batch_n_acc)
                    f1.add(predict_type, batch_f1_score)
                    nacc.add(predict_type, batch_n_acc)

        if is_main() and ((batch + 1) % cfg.log_interval == 0 or batch + 1 == batches):
            lm = loss_stat.means()
            loss_str = "loss:[" + " ".join("{}:{:.3f}".format(n.split("loss_")[-1], v) for n, v in lm.items()) + "]"
            logger = get_root_logger()
            head = (f"train-epoch[{epoch+1}]-[{batch+1}/{batches}] " + f"time:{(time.time()- end):.2f}, data_time: {data_time:.2f}, "
                    + f"{loss_str}, " + f"lr:{optimizer.param_groups[0]['lr']:.6f}, ")
            if not grec:
                m = det.means()
                logger.info(head + "".join("{}Acc:{:.2f}, ".format(MAP_DICT[i], m[MAP_DICT[i]]) for i in range(len(predictions_list))))
            else:
                mf, mn = f1.means(), nacc.means()
                logger.info(head + "".join("{}_f1: {:.2f}, ".format(MAP_DICT[i], mf[MAP_DICT[i]]) for i in range(len(predictions_list)))
                            + "".join("{}_Nacc: {:.2f}, ".format(MAP_DICT[i], mn[MAP_DICT[i]]) for i in range(len(predictions_list))))
        end = time.time()
    return loss_stat.means()
