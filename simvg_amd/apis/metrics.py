"""Metrics and per-step statistics of the outer loop.

Same numbers as the reference's `accuracy` (`simvg/apis/test.py:70-88`, Det@0.5 through mmdet's aligned
`bbox_overlaps`) and `grec_evaluate_f1_nacc` (`:91-163`, generalised-REC F1 at score 0.7 / GIoU 0.5 plus the
no-target accuracy); the values are pinned against the reference's own functions executed on seeded boxes
(`oracle/make_golden_apis.py` -> `tests/golden/apis_golden.pt`, `tests/test_apis_cpu.py`).

Built for a device that must not be stalled once per batch: metric values stay device scalars, a step's logged
scalars (5 losses + 2-4 metrics) travel between ranks as ONE packed all-reduce (`StepScalars`; the reference issues
one `reduce_mean` per value), and running means are read back only when a log line is due.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist

BRANCHES = ("decoder", "token")            # index of a prediction dict in the model's output list -> its name
MAP_DICT = dict(enumerate(BRANCHES))       # the reference's name for the same mapping


# ------------------------------------------------------------------------------------------------ boxes (xyxy)
def _area(b):
    return (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])


def bbox_overlaps_aligned(a, b, eps=1e-6):
    """IoU of row i of `a` with row i of `b`, union clamped at eps (mmdet 2.x bbox_overlaps, is_aligned=True)"""
    wh = (torch.minimum(a[..., 2:], b[..., 2:]) - torch.maximum(a[..., :2], b[..., :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (_area(a) + _area(b) - inter).clamp(min=eps)


def box_area(boxes):
    return _area(boxes)


def box_iou(a, b):
    """pairwise IoU [len(a), len(b)] and the unions"""
    wh = (torch.minimum(a[:, None, 2:], b[None, :, 2:]) - torch.maximum(a[:, None, :2], b[None, :, :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = _area(a)[:, None] + _area(b)[None, :] - inter
    return inter / union, union


def generalized_box_iou(a, b):
    """pairwise GIoU (no epsilons: the reference's in-tree copy, `simvg/apis/test.py:49-67`)"""
    assert bool((a[:, 2:] >= a[:, :2]).all()) and bool((b[:, 2:] >= b[:, :2]).all())
    iou, union = box_iou(a, b)
    hull_wh = (torch.maximum(a[:, None, 2:], b[None, :, 2:]) - torch.minimum(a[:, None, :2], b[None, :, :2])).clamp(min=0)
    hull = hull_wh[..., 0] * hull_wh[..., 1]
    return iou - (hull - union) / hull


# ------------------------------------------------------------------------------------------------ RefCOCO: Det@0.5
def accuracy(pred_bboxes, gt_bbox, pred_masks, gt_mask, is_crowd=None, device="cuda:0"):
    """-> (Det@0.5 in %, mask IoU %, mask accuracy at 5 thresholds %); the mask slots carry the reference's placeholders
    (0 and -100: its segmentation heads are outside this path) and masks are refused rather than ignored."""
    if pred_masks is not None:
        raise NotImplementedError("mask metrics need the segmentation heads, which are outside this hot path")
    if pred_bboxes is None:
        det = torch.zeros(1, device=device)
    else:
        hit = bbox_overlaps_aligned(torch.stack(list(gt_bbox)).to(device), pred_bboxes.to(device)) >= 0.5
        det = hit.float().mean()
    return det * 100.0, torch.zeros(1, device=device) * 100.0, torch.full((5,), -1.0, device=device) * 100.0


# ------------------------------------------------------------------------------------------------ GRefCOCO: F1 / N-acc
def _host_predictions(predictions):
    """per image (scores [k], boxes [k, 4]) as numpy, k may differ per image; ONE device-to-host copy when the container
    offers it (`KeptInstances.host_arrays`), else one per image"""
    if hasattr(predictions, "host_arrays"):
        return predictions.host_arrays()
    out = []
    for p in predictions:
        out.append((p["scores"].detach().reshape(-1).double().cpu().numpy(),
                    p["boxes"].detach().reshape(-1, 4).float().cpu().numpy()))
    return out


def _greedy_true_positives(giou, thresh):
    """Matches found by repeatedly taking the best remaining (prediction, target) pair while its GIoU reaches `thresh`,
    each prediction and each target used once -- what the reference's top-1 / zero-the-row-and-column loop computes."""
    if giou.size == 0:
        return 0
    n_pred, n_gt = giou.shape
    used_p, used_g = np.zeros(n_pred, bool), np.zeros(n_gt, bool)
    hits = 0
    for flat in np.argsort(-giou, axis=None, kind="stable"):
        p, g = divmod(int(flat), n_gt)
        if giou[p, g] < thresh or hits == min(n_pred, n_gt):
            break
        if not (used_p[p] or used_g[g]):
            used_p[p] = used_g[g] = True
            hits += 1
    return hits


def grec_evaluate_f1_nacc(predictions, gt_bboxes, targets, thresh_score=0.7, thresh_iou=0.5, thresh_F1=1.0, device="cuda:0"):
    """-> (share of images with F1 >= thresh_F1, no-target accuracy TP / (TP + FN)), both in %.
    An image counts its predictions with score >= thresh_score; a no-target image is right iff there is none."""
    if predictions is None:
        z = torch.tensor(0.0, device=device).float()
        return z, z.clone()
    perfect = images = 0
    no_target_right = no_target_wrong = 0
    for (scores, boxes), gts, tgt in zip(_host_predictions(predictions), gt_bboxes, targets):
        order = np.argsort(-scores, kind="stable")
        keep = order[scores[order] >= thresh_score]
        kept = torch.from_numpy(np.ascontiguousarray(boxes[keep])).float().view(-1, 4)
        pairs = list(zip(gts, tgt))
        gt_all = torch.stack([torch.as_tensor(b) for b, _ in pairs], 0).detach().cpu().float().view(-1, 4)
        giou = generalized_box_iou(kept, gt_all).numpy()
        n_pred, n_gt = kept.shape[0], gt_all.shape[0]
        if any(t["category_id"] == -1 for _, t in pairs):
            right = n_pred == 0
            no_target_right += right
            no_target_wrong += not right
            f1 = 1.0 if right else 0.0
        else:
            tp = _greedy_true_positives(giou, thresh_iou)
            f1 = 2 * tp / (2 * tp + (n_pred - tp) + (n_gt - tp))
        perfect += f1 >= thresh_F1
        images += 1
    n_acc = no_target_right / (no_target_right + no_target_wrong) if no_target_right else 0.0
    return (torch.tensor(perfect / images, device=device).float() * 100, torch.tensor(n_acc, device=device).float() * 100)


# ------------------------------------------------------------------------------------------------ statistics
class StepScalars:
    """The named device scalars one step logs.  `reduce()` averages ALL of them over the ranks with one all-reduce of a
    packed vector (a no-op outside a process group); `items()` yields them in insertion order."""

    def __init__(self):
        self._vals = OrderedDict()

    def put(self, name, value):
        self._vals[name] = value.detach().reshape(()).float()

    def reduce(self):
        if self._vals and dist.is_available() and dist.is_initialized():
            packed = torch.stack(list(self._vals.values()))
            dist.all_reduce(packed)
            packed = packed / dist.get_world_size()
            for i, k in enumerate(self._vals):
                self._vals[k] = packed[i]
        return self

    def items(self):
        return self._vals.items()


class RunningMeans:
    """per-name running mean of per-batch device scalars, read back on demand (one synchronisation per read)"""

    def __init__(self):
        self.sum, self.n = OrderedDict(), {}

    def add(self, name, value):
        value = value.detach().reshape(()).float()
        self.sum[name] = self.sum[name] + value if name in self.sum else value
        self.n[name] = self.n.get(name, 0) + 1

    def names(self):
        return list(self.sum)

    def means(self):
        if not self.sum:
            return {}
        totals = torch.stack(list(self.sum.values())).tolist()
        return {k: t / self.n[k] for k, t in zip(self.sum, totals)}


# ------------------------------------------------------------------------------------------------ batch plumbing
def unwrap(v):
    """mmcv DataContainer -> its single-GPU payload; tensors / lists pass through"""
    return v.data[0] if hasattr(v, "data") and not isinstance(v, torch.Tensor) else v


def split_gt(inputs, key="gt_bbox"):
    """`gt_bbox` arrives stacked [B, 4] (RefCOCO) or as a list of [k, 4] (GRefCOCO): always a list of per-image tensors"""
    v = inputs[key]
    return list(v.unbind(0)) if isinstance(v, torch.Tensor) else list(unwrap(v))


def score_predictions(stats, predictions, gt_bbox, img_metas, grec, device):
    """Adds the batch metrics of every branch's prediction dict to `stats` under '<metric>/<branch>' (pops
    `pred_bboxes` / `pred_masks` from the dicts, like the reference loop does)."""
    plist = predictions if isinstance(predictions, list) else [predictions]
    for branch, pred in zip(BRANCHES, plist):
        boxes, masks = pred.pop("pred_bboxes"), pred.pop("pred_masks")
        if grec:
            f1, nacc = grec_evaluate_f1_nacc(boxes, gt_bbox, [m["target"] for m in img_metas], device=device)
            stats.put("f1/" + branch, f1)
            stats.put("nacc/" + branch, nacc)
        else:
            det, _, _ = accuracy(boxes, [g.to(device) for g in gt_bbox], masks, None, device=device)
            stats.put("det/" + branch, det)
    return len(plist)
