"""Evaluation of one split (the reference's `evaluate_model`, `simvg/apis/test.py:166-293`): Det@0.5 per branch for the
single-box datasets, F1 / no-target accuracy per branch for GRefCOCO, a log line in the reference's format every
cfg.log_interval batches, and the branch-averaged pair `(d_acc, miou)` the checkpoint logic consumes (for GRefCOCO
`(f1, n_acc)`; mask IoU is 0 -- the segmentation heads are outside this path).  Metric values stay on the device; a
batch's values cross the ranks as one packed all-reduce.  The metric functions live in `apis/metrics.py`."""
import time

import torch

from ..utils import get_root_logger, is_main
from .metrics import (BRANCHES, MAP_DICT, RunningMeans, StepScalars, accuracy, bbox_overlaps_aligned, box_area,  # noqa: F401
                      box_iou, generalized_box_iou, grec_evaluate_f1_nacc, score_predictions, split_gt, unwrap)

_unwrap, _split_gt = unwrap, split_gt      # names other modules of the package import


def _eval_line(grec, means, branches, epoch, batch, batches, seconds):
    names = BRANCHES[:branches]
    if grec:
        return (f"Validate - epoch [{epoch+1}]-[{batch+1}/{batches}] time: {seconds:.2f}, "
                + "".join("{}_f1_score: {:.2f}, ".format(b, means["f1/" + b]) for b in names)
                + "".join("{}_n_acc: {:.2f}, ".format(b, means["nacc/" + b]) for b in names))
    return (f"val - epoch [{epoch+1}]-[{batch+1}/{batches}] time: {seconds:.2f}, "
            + "".join("{}Det@.5: {:.2f}, ".format(b, means["det/" + b]) for b in names))


def _branch_average(means, prefix):
    vals = [v for k, v in means.items() if k.startswith(prefix)]
    return sum(vals) / len(vals)


@torch.no_grad()
def evaluate_model(epoch, cfg, model, loader):
    from ..datasets import extract_data
    model.eval()
    device = next(model.parameters()).device
    grec = cfg["dataset"] == "GRefCOCO"
    running, branches = RunningMeans(), 1
    batches = len(loader)
    with_bbox = False
    tick = time.time()
    for batch, raw in enumerate(loader):
        raw = dict(raw)
        if "gt_mask_rle" in raw:
            raise NotImplementedError("mask evaluation is outside this hot path")
        raw.pop("is_crowd", None)
        gt_bbox = None
        if "gt_bbox" in raw:
            with_bbox = True
            gt_bbox = split_gt(raw)
            del raw["gt_bbox"]
        img_metas = unwrap(raw["img_metas"])
        predictions = model(**extract_data(raw, device), return_loss=False, rescale=False, with_bbox=with_bbox, with_mask=False)
        step = StepScalars()
        branches = score_predictions(step, predictions, gt_bbox, img_metas, grec, device)
        for name, value in step.reduce().items():
            running.add(name, value)
        if is_main() and ((batch + 1) % cfg.log_interval == 0 or batch + 1 == batches):
            get_root_logger().info(_eval_line(grec, running.means(), branches, epoch, batch, batches, time.time() - tick))
        tick = time.time()
    means = running.means()
    if grec:
        return _branch_average(means, "f1/"), _branch_average(means, "nacc/")
    return _branch_average(means, "det/"), 0
